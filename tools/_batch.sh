set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp32 or team2" 2>&1 | tail -2
for cfg in "fem3d 1024" "fem3d 256" "fem3d 128" "pwtk 1024" "kkt 512"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --no-also --matrix $1 --n $2 --dtype f32 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 f32 n=$2:', round(d['ms_per_step'],4), 'ms', round(d['value']), 'GFLOP/s frac %.3f'%d['roofline']['frac'])"
done
