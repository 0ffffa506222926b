#!/usr/bin/env python
"""EXPERIMENT driver (round 4): the output-stationary grad_value prototype of tools/exp/os_scatter.hip against the product backward.

    python tools/exp/os_scatter.py [--reps 20]

For the ring / SURVEY 8(d) sampling distributions at N = 4, S = Lq = 22223 (tools/kbench.py's generators): checks the prototype's
grad_value against `alo_hip.msda_backward` wherever no sample was routed away (far samples are only COUNTED by the prototype), then
times  scan only / scan + grad_out staging / everything  for 16 x 16 and 16 x 32 pixel tiles, next to the product kernel, which also
produces grad_loc and grad_attn.  One JSON object per line.  Not part of the library; nothing imports this.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "aloception-oss_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import alo_hip  # noqa: E402
import kbench  # noqa: E402


def build():
    so = os.path.join(HERE, "libos_scatter.so")
    src = os.path.join(HERE, "os_scatter.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.os_gv_launch.restype = ctypes.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--N", type=int, default=4)
    a = ap.parse_args()
    lib = build()
    dev = "cuda:0"
    shapes_host = (ctypes.c_int * 8)(*[v for hw in kbench.DETR_SHAPES for v in hw])
    S = sum(h * w for h, w in kbench.DETR_SHAPES)
    for kind, halo in (("encoder", (6, 6)), ("survey", (10, 7)), ("survey", (6, 6))):
        value, shapes, start, loc, attn = kbench.msda_inputs(a.N, S, kind, torch.float32)
        go = torch.randn(a.N, S, 256, device=dev)
        ref_gv = alo_hip.msda_backward(value, shapes, start, loc, attn, go)[0]
        scale = float(2.0 ** 20 / go.abs().max().item())
        gv = torch.zeros_like(ref_gv)
        counters = torch.zeros(2, dtype=torch.int64, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def launch(tile, mode):
            rc = lib.os_gv_launch(ctypes.c_void_p(loc.data_ptr()), ctypes.c_void_p(attn.data_ptr()), ctypes.c_void_p(go.data_ptr()),
                                  ctypes.c_void_p(gv.data_ptr()), ctypes.c_void_p(counters.data_ptr()), a.N, S, 8, shapes_host, tile,
                                  halo[0], halo[1], ctypes.c_float(scale), mode, stream)
            assert rc == 0, rc

        for tile, name in ((0, "16x16"), (1, "16x32")):
            counters.zero_()
            gv.zero_()
            launch(tile, 0)
            torch.cuda.synchronize()
            near, far = (int(v) for v in counters.tolist())
            err = (gv - ref_gv).abs().max().item() / ref_gv.abs().max().item()
            rec = {"experiment": "os_scatter", "locations": "ring" if kind == "encoder" else kind, "halo_px": list(halo), "tile": name,
                   "near_samples": near, "far_samples": far, "far_fraction": round(far / max(1, near + far), 5),
                   "max_err_vs_product_rel_to_largest": err,
                   "matches_product": bool(far == 0 and err < 2e-4)}
            for mode, label in ((2, "ms_scan_only"), (1, "ms_scan_and_grad_out_staging"), (0, "ms_everything")):
                rec[label] = round(kbench.time_launches(lambda: launch(tile, mode), a.reps) * 1e3, 4)
            print(json.dumps(rec), flush=True)
        t = kbench.time_launches(lambda: alo_hip.msda_backward(value, shapes, start, loc, attn, go), max(3, a.reps // 2))
        print(json.dumps({"experiment": "product msda_backward (grad_value + grad_loc + grad_attn)", "locations": "ring" if kind == "encoder" else kind,
                          "ms": round(t * 1e3, 4)}), flush=True)
        del value, loc, attn, go, gv, ref_gv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
