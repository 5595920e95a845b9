set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not child_process" 2>&1 | tail -1
for cfg in "kkt 32" "kkt 64" "kkt 96" "pwtk_shell 64" "er 64" "fem3d 64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --no-also --matrix $1 --n $2 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 n=$2:', round(d['ms_per_step'],4), d['config']['kernel_variant_resolved'], 'frac %.3f'%d['roofline']['frac'])"
done
