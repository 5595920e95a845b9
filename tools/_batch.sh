mkdir -p gpurun_out/b4
CRPSPMM_TEAM2_SHAPE=2,2,2 bash tools/prof_pmc.sh gpurun_out/b4/pmc_222 --variant 5 > gpurun_out/b4/pmc_222.txt 2>&1
CRPSPMM_TEAM2_SHAPE=0 bash tools/prof_fetch.sh gpurun_out/b4/pmc_cons --variant 5 > gpurun_out/b4/pmc_cons.txt 2>&1
cat gpurun_out/b4/pmc_222.txt gpurun_out/b4/pmc_cons.txt
