#!/bin/bash
# Round-4 refresh after the correlation epilogue change (run on the GPU box): rocprofv3 kernel stats of the detection and RAFT legs,
# the kernel micro-benchmarks, the stock-GEMM yardsticks (the correlation's shape and the long-K rate of this chip), the
# corr_gemm_exp ablations + cycle traces, and the bench line.  Outputs under gpurun_out/r04b/ (copy what is to be judged into profiles/).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args
  rm -rf /tmp/r04b_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04b_$1 -- python $ROOT/bench.py $2 > $OUT/$1_bench_under_rocprof.json 2> /tmp/r04b_$1.err
  f=$(find /tmp/r04b_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r04_$1_kernel_stats.csv
}
python $ROOT/bench.py --steps 2 --warmup 1 --raft-steps 1 --raft-warmup 1 --no-cpu-baseline --train-steps 0 --panoptic-steps 0 --micro-reps 0 --fp32-steps 0 --eager-steps 0 --no-pmc > /dev/null 2>&1
COMMON="--no-pmc --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --panoptic-steps 0"
stats detr "--no-raft --train-steps 0 --no-graph --steps 40 $COMMON"
stats raft "--steps 1 --warmup 1 --raft-steps 5 --raft-warmup 2 --train-steps 0 $COMMON"
cd $ROOT
(tools/micro/hipblaslt_corr 2>&1 | grep -v "amdgpu.ids\|algo ") > $OUT/hipblaslt_corr.txt
(tools/micro/hipblaslt_corr 1 14400 8192,2048,768 2>&1 | grep -v "amdgpu.ids\|algo ") >> $OUT/hipblaslt_corr.txt
(cd tools/exp && ./corr_gemm_exp 20 2>&1 | grep -v amdgpu.ids) > $OUT/corr_gemm_exp.txt
cp tools/exp/corr_trace_0.csv $OUT/ 2>/dev/null
FWD=msda_fused_hm,msda_fused_hm_plain,msda_fused_hm_survey,msda_fused_hm_uniform
python tools/kbench.py --which $FWD,msda_enc,msda_survey,msda_rand,msda_bwd,msda_bwd_rand,corr_build,corr_lookup,corr_lookup_bwd --reps 40 2>/dev/null | grep kernel > $OUT/kbench.txt
python bench.py > $OUT/r04_bench_line.json 2> $OUT/bench.err
tail -c 400 $OUT/r04_bench_line.json; grep corr $OUT/r04_raft_kernel_stats.csv | cut -c1-150; cat $OUT/hipblaslt_corr.txt
