set -o pipefail
mkdir -p gpurun_out/b35
CRPSPMM_TIMING=1 timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --matrix kkt240 --n 256 > gpurun_out/b35/kkt240.json 2> gpurun_out/b35/kkt240.err || { tail -5 gpurun_out/b35/kkt240.err; exit 1; }
grep "timing\|stage\|first" gpurun_out/b35/kkt240.err | head -40
python3 -c "import json;d=json.load(open('gpurun_out/b35/kkt240.json'));print(d['ms_per_step'], d['config'].get('first_exec_s'))"
nproc
