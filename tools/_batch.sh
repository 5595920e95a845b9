# scratch: the command list of the last gpurun experiment (kept so that `gpurun -- bash tools/_batch.sh` has something to run;
# the evidence runs of a round are tools/prof_r02.sh, tools/prof_pmc.sh, tools/prof_counters.sh and tools/sweep.sh)
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q
