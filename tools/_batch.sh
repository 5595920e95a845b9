set -o pipefail
mkdir -p gpurun_out/b36
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not child_process" > gpurun_out/b36/pytest.txt 2>&1 || { tail -30 gpurun_out/b36/pytest.txt; exit 1; }
tail -1 gpurun_out/b36/pytest.txt
CRPSPMM_TIMING=1 timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --matrix kkt240 --n 256 > gpurun_out/b36/kkt240.json 2> gpurun_out/b36/kkt240.err || { tail -5 gpurun_out/b36/kkt240.err; exit 1; }
grep "timing\|first" gpurun_out/b36/kkt240.err | head -40
python3 -c "import json;d=json.load(open('gpurun_out/b36/kkt240.json'));print(d['ms_per_step'], d['config'].get('first_exec_s'), d['roofline']['frac'])"
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('default:', d['ms_per_step'], d['config']['first_exec_s'], d['roofline']['frac'], d['config']['also']['ms_per_step'])"
