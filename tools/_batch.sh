set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "fem3d 256 f64" "fem3d 64 f64" "fem3d 32 f64" "kkt 256 f64" "pwtk 256 f64" "pwtk 32 f64" "pwtk 64 f64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --no-also --matrix $1 --n $2 --dtype $3 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 n=$2 $3:', round(d['ms_per_step'],4), 'frac %.3f'%d['roofline']['frac'], d['config'].get('kernel_variant_resolved'))"
done
