#!/bin/bash
# Collect PMC counters of the hand-written kernels, one counter set per rocprofv3 pass (never combined with tracing
# domains other than --kernel-trace).  Usage (on the GPU box): tools/pmc.sh "<kbench args>" ; output: gpurun_out/pmc/*.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${1:---which msda_fused --dtype bf16 --reps 3}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $ROOT/tools/kbench.py $ARGS > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python $ROOT/tools/pmc_parse.py "$f" > $OUT/pass_$i.txt 2>&1
done <<SETS
SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES
TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SETS
cat $OUT/pass_*.txt
