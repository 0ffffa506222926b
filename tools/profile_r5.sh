#!/bin/bash
# Round-5 evidence (run on the GPU box): the bench line, rocprofv3 kernel stats of the three model legs, PMC passes of the MSDA
# forward / backward on the four sampling distributions (ring / survey / trained-like / uniform), stand-alone kernel figures.
# Outputs under gpurun_out/r05/ (what is to be judged is copied into profiles/).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  rm -rf /tmp/r05_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_$1 -- python $ROOT/bench.py $2 > $OUT/$1_bench_under_rocprof.json 2> /tmp/r05_$1.err
  f=$(find /tmp/r05_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r05_$1_kernel_stats.csv
}
# MIOpen's find database warm (RAFT / training convolutions)
python $ROOT/bench.py --steps 2 --warmup 1 --raft-steps 1 --raft-warmup 1 --no-cpu-baseline --train-steps 1 --panoptic-steps 1 --micro-reps 0 --fp32-steps 0 --eager-steps 0 --trained-steps 0 --no-pmc > /dev/null 2>&1
COMMON="--no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --panoptic-steps 0 --trained-steps 0"
stats detr "--no-raft --train-steps 0 --no-graph --steps 40 $COMMON"
stats raft "--steps 1 --warmup 1 --raft-steps 5 --raft-warmup 2 --train-steps 0 $COMMON"
stats train "--steps 1 --warmup 1 --no-raft --train-steps 5 $COMMON"
rm -f $OUT/pmc_by_distribution.txt
one() {  # label, counter set, kbench selection, dtype
  rm -rf /tmp/r05_pmc
  timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/r05_pmc -- python $ROOT/tools/kbench.py --which $3 --dtype $4 --reps 3 > /tmp/r05_pmc.log 2>&1
  c=$(find /tmp/r05_pmc -name "*counter_collection.csv" | head -1)
  echo "== $1 / $2" >> $OUT/pmc_by_distribution.txt
  [ -n "$c" ] && python $ROOT/tools/pmc_parse.py $c | grep -v "^value_head" | grep -A12 "msda_" >> $OUT/pmc_by_distribution.txt
}
for which in msda_bwd msda_survey msda_trained msda_bwd_rand; do
  one "backward $which" "WRITE_SIZE" $which f32
  one "backward $which" "FETCH_SIZE" $which f32
  one "backward $which" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" $which f32
done
for kind in trained; do
  one "forward $kind" "FETCH_SIZE" msda_fused_hm_$kind bf16
  one "forward $kind" "WRITE_SIZE" msda_fused_hm_$kind bf16
  one "forward $kind" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" msda_fused_hm_$kind bf16
done
cd $ROOT
FWD=msda_fused_hm,msda_fused_hm_plain,msda_fused_hm_survey,msda_fused_hm_trained,msda_fused_hm_uniform
python tools/kbench.py --which $FWD,msda_enc,msda_survey,msda_trained,msda_rand,msda_bwd,msda_bwd_rand,corr_build,corr_lookup --reps 40 2>/dev/null | grep kernel > $OUT/kbench.txt
python bench.py > $OUT/r05_bench_line.json 2> $OUT/bench.err
tail -c 400 $OUT/r05_bench_line.json; echo; head -6 $OUT/r05_detr_kernel_stats.csv | cut -c1-160; cat $OUT/kbench.txt | cut -c1-200
