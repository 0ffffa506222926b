#!/bin/bash
# corr_lookup_kernel inside the RAFT step against the stand-alone launch: FETCH_SIZE / WRITE_SIZE / L2 hits per launch
# (one counter set per rocprofv3 run, kernel-trace only).  -> gpurun_out/pmc_lookup.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_lookup.txt
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
MODEL="python $ROOT/bench.py --steps 1 --warmup 1 --raft-steps 2 --raft-warmup 1 --train-steps 0 --no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --panoptic-steps 0 --trained-steps 0 --no-graph"
ALONE="python $ROOT/tools/kbench.py --which corr_lookup --reps 10"
$MODEL > /dev/null 2>&1   # MIOpen's find database warm
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  for where in model alone; do
    rm -rf /tmp/pl
    if [ $where = model ]; then cmd="$MODEL"; else cmd="$ALONE"; fi
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl -- $cmd > /tmp/pl.log 2>&1
    f=$(find /tmp/pl -name "*counter_collection.csv" 2>/dev/null | head -1)
    echo "== $where: $set" >> $OUT
    if [ -n "$f" ]; then python $ROOT/tools/pmc_parse.py "$f" | grep -A6 "corr_lookup_kernel" >> $OUT 2>&1; else tail -3 /tmp/pl.log >> $OUT; fi
  done
done
cat $OUT
