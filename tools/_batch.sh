set -o pipefail
bash tools/prof_r02.sh bench || exit 1
OUT=gpurun_out/r02
for cfg in "fem3d 1024 f64" "fem3d 1024 f32" "kkt 256 f64" "fem3d 256 f64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $1 --n $2 --dtype $3 > $OUT/bench_$1_n$2_$3.json 2> $OUT/bench_$1_n$2_$3.err || { tail -3 $OUT/bench_$1_n$2_$3.err; exit 1; }
  echo "$cfg: $(python3 -c "import json;d=json.load(open('$OUT/bench_$1_n$2_$3.json'));print(d['value'], d['unit'], d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])")"
done
for n in 256 128; do
  timeout -k 10 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --matrix kkt240 --n $n > $OUT/bench_kkt240_n$n.json 2> $OUT/bench_kkt240_n$n.err || { tail -5 $OUT/bench_kkt240_n$n.err; exit 1; }
  echo "kkt240 n=$n: $(python3 -c "import json;d=json.load(open('$OUT/bench_kkt240_n$n.json'));print(d['value'], d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d['config'].get('first_exec_s'), d['config'].get('hbm_in_use_GB'))")"
done
