#!/bin/bash
# PMC passes (one counter set per rocprofv3 run, kernel-trace only) over the GEMM / convolution benchmarks of this round.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_BF16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for tool in convbench gemmbench; do
    rm -rf /tmp/pmc4_${tool}_$i
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc4_${tool}_$i -- python $ROOT/tools/$tool.py > /tmp/pmc4_${tool}_$i.log 2>&1
    f=$(find /tmp/pmc4_${tool}_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python $ROOT/tools/pmc_parse.py "$f" > $OUT/${tool}_pass_$i.txt 2>&1; else echo "pass $i ($set): no output" > $OUT/${tool}_pass_$i.txt; fi
  done
done
cat $OUT/*.txt
