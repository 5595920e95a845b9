#!/bin/bash
# Two quick PMC passes (FETCH_SIZE; TCC hit/miss) of bench.py under the caller's environment.
# usage: tools/prof_fetch.sh <outdir> [bench.py args...]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/errors.log"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
