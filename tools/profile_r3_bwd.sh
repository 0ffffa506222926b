#!/bin/bash
# Round-3 PMC evidence for the MSDA backward on the three sampling distributions (ring / SURVEY 8(d) / uniform), GPU box only.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  for which in msda_bwd msda_survey msda_bwd_rand; do
    rm -rf /tmp/r03_bwd_$i
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/r03_bwd_$i -- python $ROOT/tools/kbench.py --which $which --dtype f32 --reps 3 > /tmp/r03_bwd.log 2>&1
    c=$(find /tmp/r03_bwd_$i -name "*counter_collection.csv" | head -1)
    echo "== $which / $set" >> $OUT/bwd_pmc.txt
    [ -n "$c" ] && python $ROOT/tools/pmc_parse.py $c | grep -A9 "msda_bwd_tiled_kernel" >> $OUT/bwd_pmc.txt
  done
done
cat $OUT/bwd_pmc.txt
