#!/bin/bash
# Three PMC passes of bench.py: FETCH_SIZE, L2 hits / misses, L1 -> L2 read requests (one counter group per pass).
# usage: tools/prof_fetch.sh <outdir> [bench.py args...]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 --no-also "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/errors.log"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "^==|FETCH_SIZE KB|TCC_HIT|TCC_MISS|READ_REQ" "$OUT/summary.txt"
exit 0
