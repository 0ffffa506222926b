"""Average rocprofv3 --pmc counters per kernel over the last launches of each kernel (skips warm-up launches).

    python tools/pmc_parse.py counter_collection.csv                                   -> table on stdout
    python tools/pmc_parse.py --traffic-json profiles/msda_fwd_traffic.json --kernel msda_fwd_bf16_resident_kernel \\
        --alg-bytes 324278016 --source "<how it was collected>" fetch_pass.csv write_pass.csv

The second form writes the FETCH_SIZE + WRITE_SIZE (KB in rocprofv3's CSV) of one kernel — collected in SEPARATE passes, FETCH_SIZE
needs three TCC slots and WRITE_SIZE two — as the `traffic_offline` object bench.py attaches to its roofline entry.
"""
import argparse
import collections
import csv
import json
import re


def per_kernel(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [values per dispatch]
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "alo::" not in name:
            continue
        m = re.search(r"(\w+_kernel(?:<[^(]*>)?)\(", name)
        short = (m.group(1) if m else name)[:90]
        per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def tail_mean(v, n=3):
    tail = v[-n:]
    return sum(tail) / len(tail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--traffic-json")
    ap.add_argument("--kernel", default="msda_fwd_bf16_resident_kernel")
    ap.add_argument("--alg-bytes", type=float, default=0.0)
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    if not a.traffic_json:
        for path in a.csv:
            for k, cs in per_kernel(path).items():
                print(k)
                for c, v in cs.items():
                    print(f"   {c:45s} {tail_mean(v):16.1f}   (n={len(v)})")
        return
    got = {}
    for path in a.csv:
        for k, cs in per_kernel(path).items():
            if k.startswith(a.kernel):
                for c in ("FETCH_SIZE", "WRITE_SIZE"):
                    if c in cs:
                        got[c] = tail_mean(cs[c]) * 1024.0   # rocprofv3 reports KB
    if set(got) != {"FETCH_SIZE", "WRITE_SIZE"}:
        raise SystemExit(f"need FETCH_SIZE and WRITE_SIZE of {a.kernel} in the given passes, found {sorted(got)}")
    out = {"bytes": round(got["FETCH_SIZE"] + got["WRITE_SIZE"]), "fetch_size_raw": round(got["FETCH_SIZE"]),
           "write_size": round(got["WRITE_SIZE"]), "alg_bytes": a.alg_bytes, "kernel": a.kernel,
           "ratio_to_algorithmic": round((got["FETCH_SIZE"] + got["WRITE_SIZE"]) / a.alg_bytes, 3) if a.alg_bytes else None,
           "source": a.source,
           "note": "FETCH_SIZE raw: gather pattern uncalibrated on gfx950 (lower bound; the guide's x2 applies to wide streaming reads "
                   "only); written by tools/pmc_parse.py --traffic-json from two rocprofv3 --pmc passes"}
    with open(a.traffic_json, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
