#!/bin/bash
# Round-3 evidence for the MSDA forward (run on the GPU box): rocprofv3 kernel stats of the detection bench, PMC passes of the
# resident kernel (FETCH / WRITE in separate passes, then the SQ sets), traffic json.  Outputs under gpurun_out/r03/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_stats -- python $ROOT/bench.py --no-raft --train-steps 0 --panoptic-steps 0 --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --no-graph --steps 40 > $OUT/detr_bench_under_rocprof.json 2> /tmp/r03_stats.log
f=$(find /tmp/r03_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r03_detr_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/r03_pmc_$i
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/r03_pmc_$i -- python $ROOT/tools/kbench.py --which msda_fused_hm,msda_fused_hm_plain --reps 3 > /tmp/r03_pmc_$i.log 2>&1
  c=$(find /tmp/r03_pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$c" ] && cp $c $OUT/pmc_pass_$i.csv && python $ROOT/tools/pmc_parse.py $c > $OUT/pmc_pass_$i.txt
done
python $ROOT/tools/pmc_parse.py --traffic-json $OUT/msda_fwd_traffic.json --kernel msda_fwd_bf16_resident_kernel --alg-bytes 324278016 \
  --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kbench.py --which msda_fused_hm: msda_fwd_bf16_resident_kernel, N=8, Lq=S=22223 (tools/profile_r3_fwd.sh, round 3)" \
  $OUT/pmc_pass_1.csv $OUT/pmc_pass_2.csv
cat $OUT/pmc_pass_*.txt | grep -v "^value_head" | head -80
head -12 $OUT/r03_detr_kernel_stats.csv
