set -o pipefail
mkdir -p gpurun_out/b28
export CRPSPMM_TEAM2_WAVES=16
bash tools/prof_counters.sh gpurun_out/b28/c16 \
     "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
     "TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
     "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
     "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" \
     "FETCH_SIZE" "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
     "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
     -- --no-also --matrix pwtk > gpurun_out/b28/c16.txt 2>&1
cat gpurun_out/b28/c16.txt
unset CRPSPMM_TEAM2_WAVES
bash tools/prof_counters.sh gpurun_out/b28/c8 \
     "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
     -- --no-also --matrix pwtk > gpurun_out/b28/c8.txt 2>&1
cat gpurun_out/b28/c8.txt
