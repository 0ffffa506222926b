#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp aloception-oss_amd/libalo_hotpath.so /tmp/orig.so
cp tools/exp/$1.so aloception-oss_amd/libalo_hotpath.so
python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "backward or linear_in or g3" 2>&1 | tail -3
python tools/kbench.py --which msda_bwd 2>&1 | grep kernel
cp /tmp/orig.so aloception-oss_amd/libalo_hotpath.so
