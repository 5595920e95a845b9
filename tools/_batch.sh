set -o pipefail
mkdir -p gpurun_out/b12
for mat in pwtk pwtk_shell kkt fem3d er; do
for n in 32 128 256 1024; do
for v in 0 5; do
  timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --matrix $mat --n $n --variant $v > gpurun_out/b12/bench_${mat}_${n}_$v.json 2> gpurun_out/b12/bench_${mat}_${n}_$v.err || { echo "FAILED $mat $n $v"; tail -3 gpurun_out/b12/bench_${mat}_${n}_$v.err; grep -q "Memory access fault" gpurun_out/b12/bench_${mat}_${n}_$v.err && exit 1; continue; }
  echo "$mat n=$n v$v: $(python3 -c "import json;d=json.load(open('gpurun_out/b12/bench_${mat}_${n}_$v.json'));print(d['roofline']['kernel'][-32:], 'frac %.3f'%d['roofline']['frac'])")"
done; done; done
