#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp aloception-oss_amd/libalo_hotpath.so /tmp/orig.so
for f in "$@"; do
  cp tools/exp/$f.so aloception-oss_amd/libalo_hotpath.so
  echo "== $f"; python tools/lkbench.py 2>&1 | grep fused
done
cp /tmp/orig.so aloception-oss_amd/libalo_hotpath.so
echo "== product"; python tools/lkbench.py 2>&1 | grep fused
