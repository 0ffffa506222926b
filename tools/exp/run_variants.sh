#!/bin/bash
# time kernel variants built into tools/exp/v*.so (dev only): swaps the library in the box-local copy of the repo
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp aloception-oss_amd/libalo_hotpath.so /tmp/orig.so
for f in "$@"; do
  cp tools/exp/$f.so aloception-oss_amd/libalo_hotpath.so
  echo "== $f"; python tools/kbench.py --which ${WHICH:-msda_bwd} 2>&1 | grep kernel
done
cp /tmp/orig.so aloception-oss_amd/libalo_hotpath.so
echo "== product"; python tools/kbench.py --which ${WHICH:-msda_bwd} 2>&1 | grep kernel
