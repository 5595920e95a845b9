set -o pipefail
mkdir -p gpurun_out/b26
for n in 32 64 96 128; do
  for mat in pwtk pwtk_shell fem3d; do
    timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $mat --n $n --sweep-variants > gpurun_out/b26/bench_${mat}_$n.json 2> gpurun_out/b26/bench_${mat}_$n.err || { tail -3 gpurun_out/b26/bench_${mat}_$n.err; exit 1; }
    echo "$mat n=$n: $(python3 -c "import json;d=json.load(open('gpurun_out/b26/bench_${mat}_$n.json'));print(d['roofline']['kernel'][:30], 'ms', d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])")"
    grep -i "variant" gpurun_out/b26/bench_${mat}_$n.err | tr '\n' ';'; echo
  done
done
