#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp aloception-oss_amd/libalo_hotpath.so /tmp/orig.so
for f in "$@"; do
  cp tools/exp/$f.so aloception-oss_amd/libalo_hotpath.so
  echo "== $f"; python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "head_major or fused or bench" 2>&1 | tail -1
  for i in 1 2; do python tools/kbench.py --which msda_fused_hm 2>&1 | grep fused_hm | cut -c1-110; done
done
cp /tmp/orig.so aloception-oss_amd/libalo_hotpath.so
echo "== product"; for i in 1 2; do python tools/kbench.py --which msda_fused_hm 2>&1 | grep fused_hm | cut -c1-110; done
