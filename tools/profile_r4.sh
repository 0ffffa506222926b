#!/bin/bash
# Round-4 evidence (run on the GPU box): rocprofv3 kernel stats of the three model legs of bench.py, PMC passes of the MSDA forward
# (resident and plain head-major kernel on the ring / SURVEY 8(d) / uniform sampling distributions) and of the tiled backward on the
# same three, the traffic json of the dominant kernel, the stock-GEMM yardstick and the micro-benchmarks the round's pricing rests
# on.  Outputs under gpurun_out/r04/ (copy what is to be judged into profiles/).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  rm -rf /tmp/r04_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_$1 -- python $ROOT/bench.py $2 > $OUT/$1_bench_under_rocprof.json 2> /tmp/r04_$1.err
  f=$(find /tmp/r04_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r04_$1_kernel_stats.csv
}
# MIOpen's find database warm (RAFT / training convolutions): the first process of a shape benchmarks candidates
python $ROOT/bench.py --steps 2 --warmup 1 --raft-steps 1 --raft-warmup 1 --no-cpu-baseline --train-steps 1 --panoptic-steps 1 --micro-reps 0 --fp32-steps 0 --eager-steps 0 --no-pmc > /dev/null 2>&1
COMMON="--no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --panoptic-steps 0"
stats detr "--no-raft --train-steps 0 --no-graph --steps 40 $COMMON"
stats raft "--steps 1 --warmup 1 --raft-steps 5 --raft-warmup 2 --train-steps 0 $COMMON"
stats train "--steps 1 --warmup 1 --no-raft --train-steps 5 $COMMON"
pmc() {  # tag, counter set, kbench selection, dtype
  rm -rf /tmp/r04_pmc
  timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/r04_pmc -- python $ROOT/tools/kbench.py --which $3 --dtype $4 --reps 3 > /tmp/r04_pmc.log 2>&1
  c=$(find /tmp/r04_pmc -name "*counter_collection.csv" | head -1)
  echo "== $1 / $2" >> $OUT/pmc.txt
  [ -n "$c" ] && cp $c $OUT/pmc_$1.csv && python $ROOT/tools/pmc_parse.py $c | grep -v "^value_head" >> $OUT/pmc.txt
}
rm -f $OUT/pmc.txt
FWD=msda_fused_hm,msda_fused_hm_plain,msda_fused_hm_survey,msda_fused_hm_uniform
pmc fwd_fetch "FETCH_SIZE" $FWD bf16
pmc fwd_write "WRITE_SIZE" $FWD bf16
pmc fwd_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $FWD bf16
pmc fwd_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" $FWD bf16
BWD=msda_bwd,msda_survey,msda_bwd_rand
pmc bwd_write "WRITE_SIZE" $BWD f32
pmc bwd_fetch "FETCH_SIZE" $BWD f32
pmc bwd_tcc "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" $BWD f32
python $ROOT/tools/pmc_parse.py --traffic-json $OUT/msda_fwd_traffic.json --kernel msda_fwd_bf16_resident_kernel --alg-bytes 324278016 --stream-bytes 163900000 \
  --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kbench.py --which msda_fused_hm,...: msda_fwd_bf16_resident_kernel, N=8, Lq=S=22223 (tools/profile_r4.sh, round 4; the mean covers the ring / survey / uniform launches of the pass)" \
  $OUT/pmc_fwd_fetch.csv $OUT/pmc_fwd_write.csv
cd $ROOT
(tools/micro/hipblaslt_corr 2>&1 | grep -v "amdgpu.ids\|algo ") > $OUT/hipblaslt_corr.txt
(tools/micro/lds_atomic; tools/micro/atomic_width; tools/micro/atomic_scope) 2>&1 | grep -v amdgpu.ids > $OUT/atomic_micro.txt
python tools/kbench.py --which $FWD,msda_enc,msda_survey,msda_rand,msda_bwd,msda_bwd_rand,corr_build,corr_lookup --reps 40 2>/dev/null | grep kernel > $OUT/kbench.txt
python bench.py > $OUT/r04_bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/r04_bench_line.json; head -8 $OUT/r04_raft_kernel_stats.csv | cut -c1-150; cat $OUT/hipblaslt_corr.txt
# the panoptic leg alone (configs[4] per-GPU share) under rocprofv3
stats panoptic "--steps 2 --warmup 1 --no-raft --train-steps 0 --panoptic-steps 10 --no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0"
