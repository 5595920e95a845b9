set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('default:', round(d['ms_per_step'],4), 'first_exec', round(d['config']['first_exec_s'],3), 'hbm GB', round(d['config']['hbm_in_use_GB'],2), 'frac', round(d['roofline']['frac'],3))"
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --n 32 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('n=32:', round(d['ms_per_step'],4), 'first_exec', round(d['config']['first_exec_s'],3), 'frac', round(d['roofline']['frac'],3))"
