#!/bin/bash
# kernel sequence of one eager detection step (rocprofv3 --kernel-trace), with the neighbours of every small copy kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python $ROOT/bench.py --no-raft --train-steps 0 --panoptic-steps 0 --no-cpu-baseline --micro-reps 0 --fp32-steps 0 --eager-steps 0 --no-graph --steps 3 --warmup 2 > /dev/null 2> /tmp/seq.log
f=$(find /tmp/seq -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n[:70]
names = [short(r["Kernel_Name"]) for r in rows]
# last step = from the last stem kernel on
idx = max(i for i, n in enumerate(names) if "stem_conv_pool" in n)
seq = names[idx:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[idx:]]
print(len(seq), "launches in the step")
for i, n in enumerate(seq):
    if "copy" in n.lower() or "elementwise" in n or "reduce_kernel" in n or "CatArray" in n or "fill" in n.lower():
        print(f"{i:4d} {dur[i]:6.1f} us {n:70s} after {seq[i-1][:40]:40s} before {seq[i+1][:40] if i + 1 < len(seq) else ''}")
PY
