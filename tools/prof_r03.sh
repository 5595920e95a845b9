#!/bin/bash
# Round-3 evidence run on the GPU box.  Two gpurun calls:
#   prof_r03.sh        : rocprofv3 kernel statistics of the default bench command, then the PMC passes (one counter group per
#                        pass, tools/prof_pmc.sh) for the bench workload, its irregular companion, the nlpkkt stand-in, the
#                        Queen stand-in at n = 1024 in fp64 and fp32;  tools/make_traffic_json.py turns the summaries into
#                        profiles/r03_traffic.json
#   prof_r03.sh bench  : (after r03_traffic.json is in place) the default bench line and the n = 32 / 256 / 1024 sweep
# Everything goes to gpurun_out/r03/ ; copy what is judged into profiles/.
set -o pipefail
OUT=gpurun_out/r03
mkdir -p $OUT
if [ "${1:-}" = "bench" ]; then
  timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -5 $OUT/bench_n1.err; exit 1; }
  cut -c1-2000 $OUT/bench_n1.json
  bash tools/sweep.sh 1 $OUT/sweep.jsonl > /dev/null || exit 1
  cut -c1-300 $OUT/sweep.jsonl
  for cfg in "kkt 256" "fem3d 256" "fem3d 1024"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --matrix $1 --n $2 --no-cpu-baseline --no-also --steps 50 > $OUT/bench_$1_n$2.json 2>/dev/null || exit 1
  done
  timeout -k 10 300 python bench.py --matrix fem3d --n 1024 --dtype f32 --no-cpu-baseline --steps 50 > $OUT/bench_fem3d_n1024_f32.json 2>/dev/null || exit 1
  exit 0
fi
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --host-exec 0 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1 ) || { tail -5 $OUT/stats.log; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv
bash tools/prof_pmc.sh $OUT/pmc_pwtk --no-also --host-exec 0 > $OUT/pmc_pwtk.txt 2>&1 || exit 1
bash tools/prof_pmc.sh $OUT/pmc_shell --matrix pwtk_shell --host-exec 0 > $OUT/pmc_shell.txt 2>&1 || exit 1
bash tools/prof_pmc.sh $OUT/pmc_kkt --matrix kkt --host-exec 0 > $OUT/pmc_kkt.txt 2>&1 || exit 1
bash tools/prof_pmc.sh $OUT/pmc_fem3d --matrix fem3d --n 1024 --host-exec 0 > $OUT/pmc_fem3d.txt 2>&1 || exit 1
bash tools/prof_pmc.sh $OUT/pmc_fem3d_f32 --matrix fem3d --n 1024 --dtype f32 > $OUT/pmc_fem3d_f32.txt 2>&1 || exit 1
grep -E "^==|FETCH_SIZE KB|WRITE_SIZE =" $OUT/pmc_*.txt
