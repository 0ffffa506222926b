#!/bin/bash
# Round-4 PMC passes, one rocprofv3 run per (counter set, sampling distribution): MSDA forward (resident + plain head-major kernel)
# and tiled backward on the ring / SURVEY 8(d) / uniform locations.  FETCH_SIZE and WRITE_SIZE in passes of their own (guide:
# MI355X_MICROARCH.md, HBM / rocprofv3 section).  Output: gpurun_out/r04/pmc_by_distribution.txt + the ring traffic json.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc_by_distribution.txt
one() {  # label, counter set, kbench selection, dtype, csv name to keep (optional)
  rm -rf /tmp/r04_pmc
  timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/r04_pmc -- python $ROOT/tools/kbench.py --which $3 --dtype $4 --reps 3 > /tmp/r04_pmc.log 2>&1
  c=$(find /tmp/r04_pmc -name "*counter_collection.csv" | head -1)
  echo "== $1 / $2" >> $OUT/pmc_by_distribution.txt
  if [ -n "$c" ]; then
    python $ROOT/tools/pmc_parse.py $c | grep -v "^value_head" | grep -A12 "msda_" >> $OUT/pmc_by_distribution.txt
    [ -n "$5" ] && cp $c $OUT/$5
  fi
}
for kind in ring survey uniform; do
  if [ $kind = ring ]; then sel=msda_fused_hm,msda_fused_hm_plain; else sel=msda_fused_hm_$kind; fi
  one "forward $kind" "FETCH_SIZE" $sel bf16 pmc_fwd_${kind}_fetch.csv
  one "forward $kind" "WRITE_SIZE" $sel bf16 pmc_fwd_${kind}_write.csv
  one "forward $kind" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $sel bf16
done
one "forward ring" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" msda_fused_hm,msda_fused_hm_plain bf16
for which in msda_bwd msda_survey msda_bwd_rand; do
  one "backward $which" "WRITE_SIZE" $which f32
  one "backward $which" "FETCH_SIZE" $which f32
  one "backward $which" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" $which f32
done
python $ROOT/tools/pmc_parse.py --traffic-json $OUT/msda_fwd_traffic.json --kernel msda_fwd_bf16_resident_kernel --alg-bytes 324278016 --stream-bytes 163900000 \
  --source "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kbench.py --which msda_fused_hm: msda_fwd_bf16_resident_kernel, N=8, Lq=S=22223, ring locations (tools/pmc_r4.sh, round 4)" \
  $OUT/pmc_fwd_ring_fetch.csv $OUT/pmc_fwd_ring_write.csv
cat $OUT/pmc_by_distribution.txt
cat $OUT/msda_fwd_traffic.json
