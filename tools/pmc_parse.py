"""Average rocprofv3 --pmc counters per kernel over the last launches of each kernel (skips warm-up launches)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [values per dispatch]
for r in rows:
    name = r["Kernel_Name"]
    if "alo::" not in name:
        continue
    m = re.search(r"(\w+_kernel(?:<[^(]*>)?)\(", name)
    short = (m.group(1) if m else name)[:90]
    per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in per.items():
    print(k)
    for c, v in cs.items():
        tail = v[-3:]
        print(f"   {c:45s} {sum(tail) / len(tail):16.1f}   (n={len(v)})")
