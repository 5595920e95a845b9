#!/bin/bash
# Arbitrary PMC passes of bench.py (one rocprofv3 run per quoted counter group; --pmc with --kernel-trace only).
# usage: tools/prof_counters.sh <outdir> "<group 1>" ["<group 2>" ...] -- [bench.py args...]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GROUPS_=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ $# -gt 0 ] && shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --check 0 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/errors.log"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
[ -f "$OUT/errors.log" ] && cat "$OUT/errors.log"
exit 0
