set -o pipefail
mkdir -p gpurun_out/b16
for v in 0 1 2 3; do
  if [ $v -eq 0 ]; then unset CRPSPMM_LIB_PATH; else export CRPSPMM_LIB_PATH=$GRAFT_REPO_ROOT/crp-spmm_amd/lib_exp$v/libcrpspmm_hip.so; fi
  for mat in pwtk pwtk_shell; do
    timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix $mat > gpurun_out/b16/bench_${mat}_$v.json 2> gpurun_out/b16/bench_${mat}_$v.err || { tail -3 gpurun_out/b16/bench_${mat}_$v.err; exit 1; }
    echo "cstore $v $mat: $(python3 -c "import json;d=json.load(open('gpurun_out/b16/bench_${mat}_$v.json'));print(d['roofline']['kernel'][-32:], 'frac %.3f'%d['roofline']['frac'])")"
  done
  bash tools/prof_fetch.sh gpurun_out/b16/pmc_$v --no-also > gpurun_out/b16/pmc_$v.txt 2>&1; grep -E "FETCH_SIZE KB|TCC_HIT|TCC_MISS" gpurun_out/b16/pmc_$v.txt
done
