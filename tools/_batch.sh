set -o pipefail
mkdir -p gpurun_out/b34
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not child_process" > gpurun_out/b34/pytest.txt 2>&1 || { tail -30 gpurun_out/b34/pytest.txt; exit 1; }
tail -1 gpurun_out/b34/pytest.txt
for cfg in "pwtk 32" "pwtk 64" "pwtk 48" "pwtk_shell 64" "fem3d 64" "kkt 64"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix $1 --n $2 > gpurun_out/b34/bench_$1_$2.json 2> gpurun_out/b34/bench_$1_$2.err || { tail -3 gpurun_out/b34/bench_$1_$2.err; exit 1; }
  echo "$1 n=$2: $(python3 -c "import json;d=json.load(open('gpurun_out/b34/bench_$1_$2.json'));print(d['roofline']['kernel'][-44:], 'frac %.3f'%d['roofline']['frac'])")"
done
