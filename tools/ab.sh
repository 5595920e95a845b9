#!/bin/bash
# A/B helper for the GPU box: runs bench.py for every (matrix spec) x (environment setting) and prints one line each.
#   tools/ab.sh OUTDIR "pwtk 256|fem3d 1024 f32|..." "CRPSPMM_T2_LATORDER=0|CRPSPMM_T2_LATORDER=1|CRPSPMM_LIB_PATH=other/libcrpspmm_hip.so|..."
# Matrix spec: name, n, optional dtype.  Lines go to OUTDIR/ab.txt as well; bench JSON lines to OUTDIR/*.json.
set -o pipefail
OUT=$1; mkdir -p $OUT
IFS='|' read -ra MATS <<< "$2"
IFS='|' read -ra ENVS <<< "$3"
STEPS=${STEPS:-100}
i=0
for m in "${MATS[@]}"; do
  set -- $m
  name=$1; n=$2; dt=${3:-f64}
  j=0
  for e in "${ENVS[@]}"; do
    f=$OUT/${name}_n${n}_${dt}_e${j}.json
    env $e timeout -k 10 300 python bench.py --matrix $name --n $n --dtype $dt --no-cpu-baseline --no-also --host-exec 0 --steps $STEPS > $f 2> ${f%.json}.err || { echo "FAILED: $m / $e"; tail -3 ${f%.json}.err; exit 1; }
    python3 - "$f" "$m" "$e" <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel"]
ms = k[k.rfind(",") + 1:].strip().split(" ")[0]
print("%-22s %-60s step %.4f ms  kernel %s ms  frac %.3f  first %.2f s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], ms, d["roofline"]["frac"], d["config"].get("first_exec_s") or 0))
PY
    j=$((j+1))
  done
done
