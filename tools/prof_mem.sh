#!/bin/bash
# Memory-side PMC passes (per-XCD L2 "TCC" and its fabric interface "EA"): what the L2 asks of the fabric, how long those
# requests stay outstanding (RDREQ_LEVEL / RDREQ = mean latency in TCC cycles, by Little's law), and whether the L2 stalls
# on fabric credits.  One counter group per pass (--pmc with --kernel-trace only).
# usage: tools/prof_mem.sh <outdir> <program> [args...]      e.g.  tools/prof_mem.sh gpurun_out/x python3 bench.py --no-also
# <program> must be the binary that does the GPU work ITSELF -- python3, or a compiled program: under --pmc the profiler's
# preloaded library has initialised the GPU before the program starts, so a launcher that re-execs (env, bash -c, taskset,
# numactl, a "#!/usr/bin/env python3" script) would exec out of a GPU-initialised process, which takes this pool's machine
# down.  A *.py given as the program is therefore run as "python3 file.py"; the launchers named above are refused.
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PROG=$1; shift
case "$(basename "$PROG")" in
  env|bash|sh|taskset|numactl|nice|timeout|stdbuf) echo "prof_mem.sh: '$PROG' re-execs its argument; name the interpreter (python3) or the binary itself" >&2; exit 2;;
esac
case "$PROG" in
  *.py) set -- "$PROG" "$@"; PROG=python3;;
esac
case "$PROG" in /*|python3|python) ;; *) [ -f "$ROOT/$PROG" ] && PROG="$ROOT/$PROG";; esac
ARGS=()
for a in "$@"; do case "$a" in *.py) [ -f "$ROOT/$a" ] && a="$ROOT/$a";; esac; ARGS+=("$a"); done
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum" \
           "TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum TCC_BUBBLE_sum TCC_SRC_FIFO_FULL_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -- "$PROG" "${ARGS[@]}" > "$OUT/pass$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/errors.log"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" ${PMC_FILTER:-} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
[ -f "$OUT/errors.log" ] && cat "$OUT/errors.log"
exit 0
