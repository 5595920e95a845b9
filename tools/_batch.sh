mkdir -p gpurun_out/b7
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "locality" > gpurun_out/b7/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/b7/pytest.txt
tail -5 gpurun_out/b7/pytest.txt
for mat in pwtk_shell pwtk; do
for extra in 0 16384; do
  export CRPSPMM_TEAM2_LDS_EXTRA=$extra CRPSPMM_TEAM2_SHAPE=2,2,2
  timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --matrix $mat --variant 5 > gpurun_out/b7/bench_${mat}_$extra.json 2> gpurun_out/b7/bench_${mat}_$extra.err
  echo "$mat extra $extra: $(python3 -c "import json;d=json.load(open('gpurun_out/b7/bench_${mat}_$extra.json'));print(d['roofline']['kernel'][-32:], 'frac %.3f'%d['roofline']['frac'])")"; tail -2 gpurun_out/b7/bench_${mat}_$extra.err
  bash tools/prof_fetch.sh gpurun_out/b7/pmc_${mat}_$extra --matrix $mat --variant 5 > gpurun_out/b7/pmc_${mat}_$extra.txt 2>&1; grep -E "FETCH_SIZE KB|TCC_HIT|TCC_MISS" gpurun_out/b7/pmc_${mat}_$extra.txt
done; done
