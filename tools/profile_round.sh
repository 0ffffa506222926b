#!/bin/bash
# Round profiles: rocprofv3 --kernel-trace --stats of the bench legs (steady state) -> gpurun_out/r02_*.csv.  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args
  rm -rf /tmp/prof_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python $ROOT/bench.py $2 > $OUT/r02_$1_line.json 2> /tmp/prof_$1.err
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r02_$1_kernel_stats.csv
}
# warm MIOpen's find database (first run of a shape benchmarks candidate kernels; later processes read the stored choice)
python $ROOT/bench.py --steps 2 --warmup 1 --raft-steps 1 --raft-warmup 1 --no-cpu-baseline --train-steps 1 --panoptic-steps 1 > /dev/null 2>&1
prof detr "--steps 30 --warmup 3 --no-raft --no-cpu-baseline --train-steps 0 --panoptic-steps 0"
prof raft "--steps 1 --warmup 1 --raft-steps 5 --raft-warmup 2 --no-cpu-baseline --train-steps 0 --panoptic-steps 0"
prof train "--steps 1 --warmup 1 --no-raft --no-cpu-baseline --train-steps 5 --panoptic-steps 0"
head -12 $OUT/r02_raft_kernel_stats.csv | cut -c1-160
