"""LDS-resident forward vs the plain head-major forward over pyramid sizes and batch sizes (where does the resident kernel pay?).

    python tools/exp/res_sweep.py            -> one line per (frame size, N): ms resident, ms plain, workgroups of the resident launch
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kbench  # noqa: E402
from kbench import alo_hip, DEV, time_launches  # noqa: E402


def pyramid(h, w):
    out = []
    h, w = -(-h // 8), -(-w // 8)
    for _ in range(4):
        out.append((h, w))
        h, w = -(-h // 2), -(-w // 2)
    return out


def inputs(N, shapes_list, seed=0):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    shapes = torch.tensor(shapes_list, dtype=torch.int32, device=DEV)
    shapes._alo_shapes = list(shapes_list)
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]]).to(torch.int32)
    S = sum(h * w for h, w in shapes_list)
    value = torch.randn(N, S, 8, 32, generator=gen, device=DEV).to(torch.bfloat16)
    refs = []
    for (h, w) in shapes_list:
        ys, xs = torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing="ij")
        refs.append(torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, 4, 2).contiguous()
    ang = torch.arange(8, device=DEV, dtype=torch.float32) * (2 * torch.pi / 8)
    ring = torch.stack([ang.cos(), ang.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    steps = torch.arange(1, 5, device=DEV, dtype=torch.float32)
    off = (ring[:, None, None, :] * steps[None, None, :, None]).expand(8, 4, 4, 2)
    offsets = off[None, None].expand(N, S, 8, 4, 4, 2) + (torch.rand(N, S, 8, 4, 4, 2, generator=gen, device=DEV) - 0.5)
    logits = torch.randn(N, S, 8, 16, generator=gen, device=DEV)
    return value, shapes, start, offsets.to(torch.bfloat16).contiguous(), logits.to(torch.bfloat16), ref, S


def graph_ms(fn, launches=40, replays=20):
    """GPU time per launch without the host's launch cost: `launches` launches captured in one HIP graph, replayed back to back."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(launches):
            fn()
    return time_launches(g.replay, replays) * 1e3 / launches


def main():
    import ctypes
    reps = int(os.environ.get("REPS", "200"))
    for (h, w) in [(256, 320), (384, 512), (480, 640), (512, 672), (608, 800), (800, 1333), (1024, 1024), (1200, 1600)]:
        for N in (1, 2, 4, 8):
            pyr = pyramid(h, w)
            value, shapes, start, offsets, logits, ref, S = inputs(N, pyr)
            vhm = alo_hip.value_head_major(value, torch.zeros(N, S, dtype=torch.bool, device=DEV))
            host = (ctypes.c_int32 * 8)(*[v for hw in pyr for v in hw])
            rl = alo_hip.lib().alo_msda_resident_levels(host, N, S, 8, 4, S, 0)
            tr = graph_ms(lambda: alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=True)) * 1e-3
            tp = graph_ms(lambda: alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=False)) * 1e-3
            print(json.dumps(dict(frame=[h, w], N=N, S=S, resident_levels=rl, ms_resident=round(tr * 1e3, 4), ms_plain=round(tp * 1e3, 4),
                                  ratio=round(tr / tp, 3))), flush=True)


if __name__ == "__main__":
    main()
