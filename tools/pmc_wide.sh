#!/bin/bash
# SQ / TCC counter passes over the wide MSDA backward (one counter set per rocprofv3 run, kernel-trace only).
#   tools/pmc_wide.sh [kind] [policy] -> gpurun_out/pmc_wide_<kind>_<policy>.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
KIND=${1:-trained}; POL=${2:-wide}
OUT=$ROOT/gpurun_out/pmc_wide_${KIND}_${POL}.txt
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_ATOMIC_sum"; do
  i=$((i+1))
  rm -rf /tmp/pw_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pw_$i -- python $ROOT/tools/exp/bwd_wide_run.py $KIND $POL > /tmp/pw_$i.log 2>&1
  f=$(find /tmp/pw_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  echo "== pass $i: $set" >> $OUT
  if [ -n "$f" ]; then python $ROOT/tools/pmc_parse.py "$f" >> $OUT 2>&1; else tail -5 /tmp/pw_$i.log >> $OUT; fi
done
cat $OUT
