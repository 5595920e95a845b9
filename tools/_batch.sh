set -o pipefail
mkdir -p gpurun_out/b18
for ph in 1 0; do
  export CRPSPMM_TEAM2_PHASE=$ph
  for mat in pwtk pwtk_shell; do
    timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-also --matrix $mat > gpurun_out/b18/bench_${mat}_$ph.json 2> gpurun_out/b18/bench_${mat}_$ph.err || { tail -3 gpurun_out/b18/bench_${mat}_$ph.err; exit 1; }
    echo "phase $ph $mat: $(python3 -c "import json;d=json.load(open('gpurun_out/b18/bench_${mat}_$ph.json'));print(d['roofline']['kernel'][-32:], 'frac %.3f'%d['roofline']['frac'])")"
    bash tools/prof_fetch.sh gpurun_out/b18/pmc_${mat}_$ph --no-also --matrix $mat > gpurun_out/b18/pmc_${mat}_$ph.txt 2>&1; grep -E "FETCH_SIZE KB|TCC_HIT|TCC_MISS" gpurun_out/b18/pmc_${mat}_$ph.txt
  done
done
