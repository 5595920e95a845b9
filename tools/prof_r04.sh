#!/bin/bash
# Round-4 evidence runs on the GPU box (each stage fits one gpurun call of <= 1200 s).  Everything goes to gpurun_out/r04/;
# copy what is judged into profiles/.
#   prof_r04.sh stats   : rocprofv3 kernel statistics of the default bench command
#   prof_r04.sh pmc A   : PMC passes (one counter group per pass, tools/prof_pmc.sh) for workload A in
#                         {pwtk, shell, kkt, fem3d, fem3d_f32}; "queen" = the three fetch passes + WRITE_SIZE at Queen_4147 size, fp32
#   prof_r04.sh bench   : the default bench line (with the CPU baseline), the n = 32 / 256 / 1024 sweep, the other stand-ins
#   prof_r04.sh big     : Queen_4147-size fp32 n = 1024, nlpkkt240-size width sweeps (light and 13-point coupling)
set -o pipefail
OUT=gpurun_out/r04
mkdir -p $OUT
case "${1:-}" in
stats)
  ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --host-exec 0 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1 ) || { tail -5 $OUT/stats.log; exit 1; }
  find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  head -4 $OUT/kernel_stats.csv
  ;;
pmc)
  case "$2" in
    pwtk)      bash tools/prof_pmc.sh $OUT/pmc_pwtk --no-also --host-exec 0 > $OUT/pmc_pwtk.txt 2>&1 ;;
    shell)     bash tools/prof_pmc.sh $OUT/pmc_shell --matrix pwtk_shell --host-exec 0 > $OUT/pmc_shell.txt 2>&1 ;;
    kkt)       bash tools/prof_pmc.sh $OUT/pmc_kkt --matrix kkt --host-exec 0 > $OUT/pmc_kkt.txt 2>&1 ;;
    fem3d)     bash tools/prof_pmc.sh $OUT/pmc_fem3d --matrix fem3d --n 1024 --host-exec 0 > $OUT/pmc_fem3d.txt 2>&1 ;;
    fem3d_f32) bash tools/prof_pmc.sh $OUT/pmc_fem3d_f32 --matrix fem3d --n 1024 --dtype f32 > $OUT/pmc_fem3d_f32.txt 2>&1 ;;
    queen)
      mkdir -p $OUT/pmc_queen_f32
      cd /tmp && export TMPDIR=/tmp
      i=0
      for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
        i=$((i+1))
        rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_queen_f32/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --matrix fem3d_queen --n 1024 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --check 0 > $GRAFT_REPO_ROOT/$OUT/pmc_queen_f32/pass$i.log 2>&1 || echo "pass $i failed"
      done
      cd $GRAFT_REPO_ROOT
      python3 tools/pmc_summary.py $OUT/pmc_queen_f32 > $OUT/pmc_queen_f32.txt 2>&1 ;;
  esac
  grep -E "^==|FETCH_SIZE KB|WRITE_SIZE|TCC_HIT|TCC_MISS" $OUT/pmc_$2*.txt | cut -c1-200
  ;;
bench)
  timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -5 $OUT/bench_n1.err; exit 1; }
  cut -c1-1500 $OUT/bench_n1.json
  bash tools/sweep.sh 1 $OUT/sweep.jsonl > /dev/null || exit 1
  for cfg in "kkt 256" "kkt 128" "kkt 64" "kkt 32" "fem3d 256" "fem3d 1024" "fem3d 128" "pwtk 128" "pwtk 64" "pwtk_shell 128"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --matrix $1 --n $2 --no-cpu-baseline --no-also --host-exec 0 --steps 50 > $OUT/bench_$1_n$2.json 2>/dev/null || exit 1
  done
  timeout -k 10 300 python bench.py --matrix fem3d --n 1024 --dtype f32 --no-cpu-baseline --steps 50 > $OUT/bench_fem3d_n1024_f32.json 2>/dev/null || exit 1
  for f in $OUT/sweep.jsonl $OUT/bench_*_n*.json; do python3 -c "
import json,sys
for l in open('$f'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-70s %.4f ms  frac %.3f' % (d['config']['workload'][:70], d['ms_per_step'], d['roofline']['frac']))
"; done
  ;;
big)
  timeout -k 10 500 python bench.py --matrix fem3d_queen --n 1024 --dtype f32 --no-cpu-baseline --steps 20 > $OUT/bench_queen_size_f32_n1024.json 2> $OUT/bench_queen.err || { tail -3 $OUT/bench_queen.err; exit 1; }
  cut -c1-400 $OUT/bench_queen_size_f32_n1024.json
  CRPSPMM_TIMING=1 timeout -k 10 1000 python tools/width_sweep.py --matrix ${2:-kkt240} --out $OUT/${2:-kkt240}_width_sweep.jsonl 2> $OUT/${2:-kkt240}_width_sweep.err || { tail -5 $OUT/${2:-kkt240}_width_sweep.err; exit 1; }
  cat $OUT/${2:-kkt240}_width_sweep.jsonl | cut -c1-600
  ;;
*) echo "usage: prof_r04.sh stats | pmc <workload> | bench | big [kkt240|kkt240d]"; exit 2;;
esac
