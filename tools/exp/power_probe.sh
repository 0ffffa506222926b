#!/bin/bash
# sample clocks / power while one kernel loops (dev only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import torch" 2>/dev/null
python tools/kbench.py --which ${WHICH:-corr_build} --reps ${REPS:-8000} > /tmp/kb.log 2>&1 &
KPID=$!
while kill -0 $KPID 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 2
done
tail -1 /tmp/kb.log | cut -c1-120
