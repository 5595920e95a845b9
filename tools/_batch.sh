set -o pipefail
OUT=gpurun_out/r02; mkdir -p $OUT
for cfg in "fem3d 1024 f64" "fem3d 1024 f32" "fem3d 256 f64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $1 --n $2 --dtype $3 > $OUT/bench_$1_n$2_$3.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('$OUT/bench_$1_n$2_$3.json'));print('$cfg:', round(d['ms_per_step'],4), round(d['value']), 'frac %.3f'%d['roofline']['frac'])"
done
