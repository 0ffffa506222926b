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
    ap.add_argument("--stream-bytes", type=float, default=0.0, help="bytes per launch the kernel reads as coalesced streams (counted at half by FETCH_SIZE)")
    a = ap.parse_args()
    if not a.traffic_json:
        for path in a.csv:
            for k, cs in per_kernel(path).items():
                print(k)
                for c, v in cs.items():
                    print(f"   {c:45s} {tail_mean(v):16.1f}   (n={len(v)})")
        return
    out = traffic(a.csv, a.kernel, a.alg_bytes, a.stream_bytes, a.source)
    with open(a.traffic_json, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


# Calibration of the two counters on gfx950 in the access shapes of the MSDA forward (tools/micro/fetch_calib.hip under
# tools/exp/fetch_calib.sh, every kernel moving exactly 1 GiB, far beyond the Infinity Cache; round 4):
#   FETCH_SIZE  coalesced streams, 16 or 8 bytes per lane            counted 0.500 of the bytes read (128-byte requests tallied at 64)
#               64-byte rows (4 lanes x 16 B) in a scattered order    counted 1.000
#               128-byte rows (8 lanes x 16 B) in a scattered order   counted 0.500
#   WRITE_SIZE  coalesced 16-byte stores                              counted 1.000
# i.e. FETCH_SIZE = read requests x 64 B with requests of 64 OR 128 bytes: the guide's "double it" holds for everything that
# coalesces to 128-byte requests.  A kernel that mixes known streams with 64-byte row gathers is corrected as
#   fetch >= streams + (raw - streams / 2)           (every gather request 64 bytes wide: counted in full)
#   fetch <= streams + 2 (raw - streams / 2)         (every gather request 128 bytes wide: two neighbouring rows in one request)
def traffic(csv_paths, kernel, alg_bytes, stream_bytes=0.0, source=""):
    """FETCH_SIZE + WRITE_SIZE per launch of ``kernel`` from rocprofv3 --pmc passes (the two counters in SEPARATE passes), raw and
    corrected with the calibration above.  ``stream_bytes`` = the bytes per launch the kernel reads as coalesced streams (known from
    its algorithm); the rest of FETCH_SIZE is attributed to its row gathers."""
    got = {}
    for path in csv_paths:
        for k, cs in per_kernel(path).items():
            if k.startswith(kernel):
                for c in ("FETCH_SIZE", "WRITE_SIZE"):
                    if c in cs:
                        got[c] = tail_mean(cs[c]) * 1024.0   # rocprofv3 reports KB
    if set(got) != {"FETCH_SIZE", "WRITE_SIZE"}:
        raise SystemExit(f"need FETCH_SIZE and WRITE_SIZE of {kernel} in the given passes, found {sorted(got)}")
    raw = got["FETCH_SIZE"] + got["WRITE_SIZE"]
    gathers_counted = max(got["FETCH_SIZE"] - 0.5 * stream_bytes, 0.0)
    lower = stream_bytes + gathers_counted + got["WRITE_SIZE"]
    upper = stream_bytes + 2.0 * gathers_counted + got["WRITE_SIZE"]
    return {"bytes": round(lower), "bytes_upper": round(upper), "bytes_raw": round(raw),
            "fetch_size_raw": round(got["FETCH_SIZE"]), "write_size": round(got["WRITE_SIZE"]), "stream_bytes": stream_bytes,
            "alg_bytes": alg_bytes, "kernel": kernel,
            "ratio_to_algorithmic": round(lower / alg_bytes, 3) if alg_bytes else None,
            "ratio_to_algorithmic_upper": round(upper / alg_bytes, 3) if alg_bytes else None,
            "ratio_raw": round(raw / alg_bytes, 3) if alg_bytes else None,
            "source": source,
            "note": "bytes = WRITE_SIZE + FETCH_SIZE corrected with the gfx950 calibration of tools/micro/fetch_calib (streams are counted "
                    "at half: + stream_bytes / 2; 64-byte row gathers in full): a lower bound, bytes_upper assumes every gather request "
                    "was 128 bytes wide; Infinity-Cache hits are counted as fetches (guide), so both bound the traffic beyond the L2, not "
                    "HBM alone"}


def traffic_sum(csv_paths, kernels, alg_bytes, stream_bytes=None, source=""):
    """The same for an ENTRY POINT made of several kernels (alo_corr_build: magnitude, split, pooling and the two GEMM launches):
    FETCH_SIZE / WRITE_SIZE summed over every dispatch of every kernel whose name starts with one of ``kernels``, divided by the
    number of dispatches of ``kernels[0]`` (one per call of the entry point).  ``stream_bytes`` None = every read of these kernels
    is a coalesced stream of 8 / 16 bytes per lane (counted at half by FETCH_SIZE on gfx950): fetch = 2 x raw, no upper bound."""
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    calls = {}
    for path in csv_paths:
        for k, cs in per_kernel(path).items():
            if any(k.startswith(pref) for pref in kernels):
                for c in tot:
                    if c in cs:
                        tot[c] += sum(cs[c]) * 1024.0
                        if k.startswith(kernels[0]):
                            calls[c] = calls.get(c, 0) + len(cs[c])
    if set(calls) != {"FETCH_SIZE", "WRITE_SIZE"} or not all(calls.values()):
        raise SystemExit(f"need FETCH_SIZE and WRITE_SIZE of {kernels} in the given passes, found {sorted(calls)}")
    fetch, write = tot["FETCH_SIZE"] / calls["FETCH_SIZE"], tot["WRITE_SIZE"] / calls["WRITE_SIZE"]
    if stream_bytes is None:
        lower = upper = 2.0 * fetch + write
    else:
        gathers = max(fetch - 0.5 * stream_bytes, 0.0)
        lower, upper = stream_bytes + gathers + write, stream_bytes + 2.0 * gathers + write
    return {"bytes": round(lower), "bytes_upper": round(upper), "bytes_raw": round(fetch + write), "fetch_size_raw": round(fetch),
            "write_size": round(write), "stream_bytes": stream_bytes, "alg_bytes": alg_bytes, "kernels": list(kernels),
            "calls_averaged": calls["FETCH_SIZE"],
            "ratio_to_algorithmic": round(lower / alg_bytes, 3) if alg_bytes else None,
            "ratio_to_algorithmic_upper": round(upper / alg_bytes, 3) if alg_bytes else None,
            "write_ratio_to_algorithmic_writes": None, "source": source,
            "note": "per call of the entry point, summed over its kernels; FETCH_SIZE corrected with the gfx950 calibration of "
                    "tools/micro/fetch_calib (coalesced streams are counted at half); Infinity-Cache hits count as fetches"}


if __name__ == "__main__":
    main()
