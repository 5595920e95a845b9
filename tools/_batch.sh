set -o pipefail
bash tools/prof_r02.sh bench || exit 1
OUT=gpurun_out/r02
for v in 0 3 2 1; do
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix kkt --n 32 --variant $v 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('kkt n=32 variant $v:', d['ms_per_step'], d['config']['kernel_variant_resolved'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.txt; [ $rc -eq 0 ] || exit 1
