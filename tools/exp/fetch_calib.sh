#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/micro/fetch_calib's kernels (each moves exactly 1 GiB): counted bytes / actual bytes per access shape.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- $ROOT/tools/micro/fetch_calib > /dev/null 2>&1
  f=$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" <<'PY'
import csv, sys, collections
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
for k, v in per.items():
    m = sum(v[-2:]) / len(v[-2:])
    print(f"{sys.argv[2]:10s} {k:40s} counted {m / 2**30:6.3f} GiB of 1 GiB moved  -> ratio {m / 2**30:5.3f}")
PY
done
