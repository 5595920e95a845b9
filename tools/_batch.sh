set -o pipefail
mkdir -p gpurun_out/b25
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team2 or locality or fp32" > gpurun_out/b25/pytest.txt 2>&1 || { tail -20 gpurun_out/b25/pytest.txt; exit 1; }
tail -1 gpurun_out/b25/pytest.txt
for cfg in "fem3d 256" "kkt 256" "pwtk_shell 256" "fem3d 1024" "er 256"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $1 --n $2 > gpurun_out/b25/bench_$1_$2.json 2> gpurun_out/b25/bench_$1_$2.err || { tail -3 gpurun_out/b25/bench_$1_$2.err; exit 1; }
  echo "$1 n=$2: $(python3 -c "import json;d=json.load(open('gpurun_out/b25/bench_$1_$2.json'));print(d['roofline']['kernel'][-40:], 'frac %.3f'%d['roofline']['frac'], 'first_exec', d['config'].get('first_exec_s'))")"
done
timeout -k 10 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --matrix kkt240 --n 256 > gpurun_out/b25/bench_kkt240.json 2> gpurun_out/b25/bench_kkt240.err || { tail -5 gpurun_out/b25/bench_kkt240.err; exit 1; }
echo "kkt240: $(python3 -c "import json;d=json.load(open('gpurun_out/b25/bench_kkt240.json'));print(d['roofline']['kernel'][-60:], 'frac %.3f'%d['roofline']['frac'], 'first_exec', d['config'].get('first_exec_s'))")"
