set -o pipefail
mkdir -p gpurun_out/b21
for ph in 1 0; do
  export CRPSPMM_TEAM2_PHASE=$ph
  for cfg in "kkt 256" "fem3d 256" "fem3d 1024" "pwtk 128" "pwtk 512" "pwtk 1024" "pwtk_shell 1024"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-also --matrix $1 --n $2 > gpurun_out/b21/bench_$1_$2_$ph.json 2> gpurun_out/b21/bench_$1_$2_$ph.err || { tail -3 gpurun_out/b21/bench_$1_$2_$ph.err; exit 1; }
    echo "phase $ph $1 n=$2: $(python3 -c "import json;d=json.load(open('gpurun_out/b21/bench_$1_$2_$ph.json'));print(d['roofline']['kernel'][-40:], 'frac %.3f'%d['roofline']['frac'])")"
  done
done
