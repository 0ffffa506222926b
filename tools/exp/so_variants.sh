#!/bin/bash
# A/B of library builds: for every aloception-oss_amd/libalo_hotpath_<tag>.so run the given kbench selection with that build in place
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp aloception-oss_amd/libalo_hotpath.so /tmp/base.so
for f in aloception-oss_amd/libalo_hotpath_*.so; do
  tag=${f##*hotpath_}; tag=${tag%.so}
  cp $f aloception-oss_amd/libalo_hotpath.so
  echo "== $tag"; python tools/kbench.py --which ${1:-msda_fused_hm} --reps ${REPS:-30} 2>&1 | grep "\"kernel\""
done
cp /tmp/base.so aloception-oss_amd/libalo_hotpath.so
