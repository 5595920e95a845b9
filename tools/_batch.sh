set -o pipefail
mkdir -p gpurun_out/b32
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not child_process" > gpurun_out/b32/pytest.txt 2>&1 || { tail -30 gpurun_out/b32/pytest.txt; exit 1; }
tail -1 gpurun_out/b32/pytest.txt
for cfg in "pwtk 32" "pwtk_shell 32" "fem3d 32" "pwtk 24"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix $1 --n $2 > gpurun_out/b32/bench_$1_$2.json 2> gpurun_out/b32/bench_$1_$2.err || { tail -3 gpurun_out/b32/bench_$1_$2.err; exit 1; }
  echo "$1 n=$2: $(python3 -c "import json;d=json.load(open('gpurun_out/b32/bench_$1_$2.json'));print(d['roofline']['kernel'][-44:], 'frac %.3f'%d['roofline']['frac'])")"
done
CRPSPMM_NARROW_MAX=64 timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix pwtk --n 64 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('n=64 narrow:', d['roofline']['kernel'][-44:], 'frac %.3f'%d['roofline']['frac'])"
