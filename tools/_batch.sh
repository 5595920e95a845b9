set -o pipefail
mkdir -p gpurun_out/b24
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team2 or locality or fp32" > gpurun_out/b24/pytest.txt 2>&1 || { tail -20 gpurun_out/b24/pytest.txt; exit 1; }
tail -2 gpurun_out/b24/pytest.txt
for cl in 1 2; do
  export CRPSPMM_TEAM2_CLUSTER=$cl
  for cfg in "fem3d 256" "kkt 256" "pwtk_shell 256" "pwtk 256"; do
    set -- $cfg
    if [ $cl = 2 ] && [ $1 != kkt ] && [ $1 != pwtk ]; then continue; fi
    timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $1 --n $2 > gpurun_out/b24/bench_$1_$2_$cl.json 2> gpurun_out/b24/bench_$1_$2_$cl.err || { tail -3 gpurun_out/b24/bench_$1_$2_$cl.err; exit 1; }
    echo "cluster $cl $1 n=$2: $(python3 -c "import json;d=json.load(open('gpurun_out/b24/bench_$1_$2_$cl.json'));print(d['roofline']['kernel'][-40:], 'frac %.3f'%d['roofline']['frac'])")"
  done
done
