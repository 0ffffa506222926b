#!/bin/bash
# Collect PMC counters of the hand-written kernels, one counter set per rocprofv3 pass (never combined with tracing
# domains other than --kernel-trace).  Run on the GPU box:
#   tools/pmc.sh "<kbench args>" "SET 1 counters;SET 2 counters;..."      -> gpurun_out/pmc/pass_<i>.txt
# TA_*/TCP_*/TD_* derived counters hung on this pool (each pass is capped at 150 s and skipped if it does not finish).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${1:---which msda_fused_hm --dtype bf16 --reps 3}
SETS=${2:-FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_BF16}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra LIST <<< "$SETS"
for set in "${LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $ROOT/tools/kbench.py $ARGS > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $ROOT/tools/pmc_parse.py "$f" > $OUT/pass_$i.txt 2>&1; else echo "pass $i ($set): no output" > $OUT/pass_$i.txt; fi
done
cat $OUT/pass_*.txt
