"""Static instruction census of the main loop of msda_fwd_bf16_resident_kernel (device assembly from hipcc -S), per executed path.

    python tools/exp/isa_census.py            # compiles aloception-oss_amd/csrc/msda.hip to /tmp/msda.s and prints the table

The loop body holds stage 1 (prologue + four descriptors per lane), then EITHER the resident path (levels 0-1 through the buffer
path, levels 2-3 from LDS) OR the all-buffer path (host hint disagrees with the device metadata), then the store.  Blocks are told
apart by their matrix / LDS / buffer instruction counts, not by label numbers.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def census(lines):
    c = collections.Counter()
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith((".", ";")) or ln.endswith(":"):
            continue
        op = ln.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["v_perm"] += op.startswith("v_perm")
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def main():
    asm = "/tmp/msda.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only",
                           "-S", "-o", asm, os.path.join(ROOT, "aloception-oss_amd", "csrc", "msda.hip")], stderr=subprocess.DEVNULL)
    L = open(asm).read().split("\n")
    start = next(i for i, ln in enumerate(L) if re.match(r"^_ZN3alo\S*msda_fwd_bf16_resident_kernel\S*:", ln))
    end = next(i for i in range(start, len(L)) if ".Lfunc_end" in L[i])
    body = L[start:end]
    marks = [i for i, ln in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", ln)] + [len(body)]
    blocks = [(a, b, census(body[a:b])) for a, b in zip(marks, marks[1:])]
    heavy = [(a, b, c) for a, b, c in blocks if c["valu"] >= 100 or c["mfma"]]
    print(f"{'block (asm lines)':22s} {'VALU':>5s} {'v_perm':>6s} {'MFMA':>5s} {'LDS':>4s} {'VMEM':>5s} {'SALU':>5s} {'waitcnt':>7s}")
    for a, b, c in heavy:
        kind = ("all-buffer path (not taken when the hint matches)" if c["mfma"] and c["vmem"] >= 60 else
                "resident path: 2 buffer groups + 8 LDS samples" if c["mfma"] else "stage 1 / other")
        print(f"{a:6d}-{b:<6d}         {c['valu']:5d} {c['v_perm']:6d} {c['mfma']:5d} {c['lds']:4d} {c['vmem']:5d} {c['salu']:5d} {c['s_waitcnt']:7d}   {kind}")


if __name__ == "__main__":
    sys.exit(main())
