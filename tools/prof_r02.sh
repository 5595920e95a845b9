#!/bin/bash
# Round-2 evidence run on the GPU box: full GPU test suite, the default bench line, the rocprofv3 kernel statistics of
# the same command, and the PMC passes (one counter group per pass) for both pwtk-class stand-ins.
# Two gpurun calls: `prof_r02.sh` (tests, rocprofv3 stats, PMC), then -- after profiles/r02_traffic.json has been
# refreshed from its PMC summary -- `prof_r02.sh bench` (the default bench line and the n = 32 / 256 / 1024 sweep).
# Everything goes to gpurun_out/r02/ ; copy what is judged into profiles/.
set -o pipefail
OUT=gpurun_out/r02
mkdir -p $OUT
if [ "${1:-}" = "bench" ]; then
  # second call, after profiles/r02_traffic.json has been refreshed from the PMC passes of the first
  timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -5 $OUT/bench_n1.err; exit 1; }
  cut -c1-1800 $OUT/bench_n1.json
  bash tools/sweep.sh 1 $OUT/sweep.jsonl > /dev/null || exit 1
  cut -c1-400 $OUT/sweep.jsonl
  exit 0
fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; rc=$?; tail -4 $OUT/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1 ) || { tail -5 $OUT/stats.log; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -5 $OUT/kernel_stats.csv
bash tools/prof_pmc.sh $OUT/pmc_pwtk --no-also > $OUT/pmc_pwtk.txt 2>&1 || exit 1
bash tools/prof_pmc.sh $OUT/pmc_shell --matrix pwtk_shell > $OUT/pmc_shell.txt 2>&1 || exit 1
grep -E "^==|FETCH_SIZE KB|WRITE_SIZE =|TCC_HIT|TCC_MISS" $OUT/pmc_pwtk.txt $OUT/pmc_shell.txt
