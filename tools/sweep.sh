#!/bin/bash
# The report BASELINE.json asks for: pwtk stand-in at n in {32, 256, 1024} on 1/2/4/8 GPUs of this node,
# one bench.py JSON line per configuration (GFLOP/s in "value", roofline fraction in "roofline.frac"),
# appended to one file (default gpurun_out/sweep.jsonl; copy it to profiles/rNN_sweep.jsonl to track it).
# usage: tools/sweep.sh [max_gpus] [outfile]     (needs that many GPUs; one rank per GPU over RCCL)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
MAXG=${1:-$(python3 -c "import torch; print(torch.cuda.device_count())")}
OUT=${2:-$ROOT/gpurun_out/sweep.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for n in 32 256 1024; do
  for g in 1 2 4 8; do
    [ "$g" -gt "$MAXG" ] && continue
    if [ "$g" -eq 1 ]; then
      python3 "$ROOT/bench.py" --n $n --no-cpu-baseline --no-also 2>/dev/null | tail -1 | tee -a "$OUT"
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29600 + g)) \
        "$ROOT/bench.py" --gpus $g --n $n --no-cpu-baseline --no-also 2>/dev/null | tail -1 | tee -a "$OUT"
    fi
  done
done
