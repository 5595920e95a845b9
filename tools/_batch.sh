set -o pipefail
mkdir -p gpurun_out/b10
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team2 or locality" > gpurun_out/b10/pytest.txt 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/b10/pytest.txt
tail -8 gpurun_out/b10/pytest.txt
[ $rc -eq 0 ] || exit 1
for mat in pwtk pwtk_shell; do
  export CRPSPMM_TEAM2_SHAPE=2,2,2
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --matrix $mat --variant 5 > gpurun_out/b10/bench_${mat}.json 2> gpurun_out/b10/bench_${mat}.err || { tail -3 gpurun_out/b10/bench_${mat}.err; exit 1; }
  echo "$mat: $(python3 -c "import json;d=json.load(open('gpurun_out/b10/bench_${mat}.json'));print(d['roofline']['kernel'][-32:], 'frac %.3f'%d['roofline']['frac'])")"
done
