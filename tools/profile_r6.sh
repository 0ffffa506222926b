#!/bin/bash
# Round-6 evidence (run on the GPU box): the bench line + sidecar, rocprofv3 kernel stats of the three model legs, counter passes of
# the wide MSDA backward on the four sampling distributions (and of the 4x4 tiled kernel beside it), stand-alone kernel figures.
# Outputs under gpurun_out/r06/ (what is to be judged is copied into profiles/).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  rm -rf /tmp/r06_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_$1 -- python $ROOT/bench.py $2 --detail-out /tmp/r06_$1_detail.json > $OUT/$1_bench_under_rocprof.json 2> /tmp/r06_$1.err
  f=$(find /tmp/r06_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r06_$1_kernel_stats.csv
}
# MIOpen's find database warm (RAFT / training convolutions)
python $ROOT/bench.py --steps 2 --warmup 1 --raft-steps 1 --raft-warmup 1 --no-cpu-baseline --train-steps 1 --panoptic-steps 1 --micro-reps 0 --fp32-steps 0 --eager-steps 0 --trained-steps 0 --no-pmc --detail-out /tmp/r06_warm.json > /dev/null 2>&1
COMMON="--no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --panoptic-steps 0 --trained-steps 0"
stats detr "--no-raft --train-steps 0 --no-graph --steps 40 $COMMON"
stats raft "--steps 1 --warmup 1 --raft-steps 5 --raft-warmup 2 --train-steps 0 $COMMON"
stats train "--steps 1 --warmup 1 --no-raft --train-steps 5 $COMMON"
rm -f $OUT/pmc_bwd_by_distribution.txt
one() {  # label, counter set, kind, policy
  rm -rf /tmp/r06_pmc
  timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/r06_pmc -- python $ROOT/tools/exp/bwd_wide_run.py $3 $4 4 5 > /tmp/r06_pmc.log 2>&1
  c=$(find /tmp/r06_pmc -name "*counter_collection.csv" | head -1)
  echo "== $1 / $2" >> $OUT/pmc_bwd_by_distribution.txt
  [ -n "$c" ] && python $ROOT/tools/pmc_parse.py $c | grep -A12 "msda_bwd" >> $OUT/pmc_bwd_by_distribution.txt
}
for kind in ring survey trained uniform; do
  one "wide $kind" "WRITE_SIZE TCC_ATOMIC_sum" $kind wide
  one "wide $kind" "FETCH_SIZE" $kind wide
done
for kind in ring trained; do
  one "wide $kind" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" $kind wide
  one "wide $kind" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE" $kind wide
  one "tiled-4x4 $kind" "WRITE_SIZE TCC_ATOMIC_sum" $kind tiled
  one "tiled-4x4 $kind" "FETCH_SIZE" $kind tiled
done
cd $ROOT
FWD=msda_fused_hm,msda_fused_hm_plain,msda_fused_hm_survey,msda_fused_hm_trained,msda_fused_hm_uniform
python tools/kbench.py --which $FWD,msda_enc,msda_rand,msda_bwd_all,msda_bwd_bf16,corr_build,corr_lookup --reps 40 2>/dev/null | grep kernel > $OUT/kbench.txt
ALO_MSDA_BWD=tiled python tools/kbench.py --which msda_bwd_all --reps 40 2>/dev/null | grep kernel | sed 's/msda_bwd\[/msda_bwd_tiled4x4[/' >> $OUT/kbench.txt
for d in 32 64; do for dt in f32 bf16; do python tools/exp/bwd_wide_check.py --N 4 --D $d --dtype $dt --kinds ring,trained 2>/dev/null | grep kind | sed "s/^/D=$d /"; done; done > $OUT/bwd_wide_check.txt
python tools/exp/lookup_mall_probe.py 2>/dev/null | grep flush > $OUT/lookup_mall_probe.txt
python tools/exp/fwd_f32_hm_probe.py 8 2>/dev/null | grep kind > $OUT/fwd_f32_hm_probe.txt
python bench.py > $OUT/r06_bench_line.json 2> $OUT/bench.err
cp $ROOT/bench_detail.json $OUT/r06_bench_detail.json
tail -c 600 $OUT/r06_bench_line.json; echo; head -6 $OUT/r06_train_kernel_stats.csv | cut -c1-160; cat $OUT/kbench.txt | cut -c1-200
