#!/bin/bash
# A/B helper for the GPU box: bench.py per (matrix, n) with each kernel variant forced; one line each.
#   VARIANTS="3 5" tools/ab_variants.sh OUTDIR "pwtk 64|fem3d 96|..."        (default VARIANTS="0 3 5")
set -o pipefail
OUT=$1; mkdir -p $OUT
IFS='|' read -ra MATS <<< "$2"
STEPS=${STEPS:-100}
for m in "${MATS[@]}"; do
  set -- $m
  name=$1; n=$2
  for v in ${VARIANTS:-0 3 5}; do
    f=$OUT/${name}_n${n}_v${v}.json
    timeout -k 10 300 python bench.py --matrix $name --n $n --variant $v --no-cpu-baseline --no-also --host-exec 0 --steps $STEPS > $f 2> ${f%.json}.err || { echo "FAILED: $m / variant $v"; tail -3 ${f%.json}.err; continue; }
    python3 - "$f" "$m" "$v" <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel"]
ms = k[k.rfind(",") + 1:].strip().split(" ")[0]
print("%-18s variant %s -> %-12s step %.4f ms  kernel %s ms  frac %.3f" % (sys.argv[2], sys.argv[3], d["config"]["kernel_variant_resolved"], d["ms_per_step"], ms, d["roofline"]["frac"]))
PY
  done
done
