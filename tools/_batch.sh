set -o pipefail
mkdir -p gpurun_out/b42
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not child_process" > gpurun_out/b42/pytest.txt 2>&1 || { tail -30 gpurun_out/b42/pytest.txt; exit 1; }
tail -1 gpurun_out/b42/pytest.txt
for cfg in "pwtk 64" "pwtk 48" "pwtk_shell 64" "fem3d 64" "pwtk 96" "kkt 64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix $1 --n $2 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 n=$2:', d['roofline']['kernel'][-44:], 'frac %.3f'%d['roofline']['frac'], d['config']['kernel_variant_resolved'])"
done
