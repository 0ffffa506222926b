#!/usr/bin/env python
"""Kernel-level micro-benchmarks of the hot path at BASELINE.json sizes (HIP events on the launch stream).

    python tools/kbench.py [--which msda_enc,msda_dec,msda_bwd,corr_build,corr_lookup] [--reps 20] [--dtype f32|bf16]

Prints one JSON object per kernel: average launch time, algorithmic bytes / flops (SURVEY.md section 8d formulas),
achieved GB/s or TFLOP/s.  Tuning knobs are environment variables of the library (ALO_MSDA_FWD_WAVES, ALO_MSDA_ITERS).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aloception-oss_amd"))
import alo_hip  # noqa: E402

DETR_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"


def time_launches(fn, reps, warmup=None):
    # the chip leaves its idle clocks only after a few milliseconds of work: a handful of warm-up launches times the ramp, not the
    # kernel (0.21 vs 0.18 ms for the same MSDA forward).  Warm up for >= 0.1 s of launches.
    if warmup is None:
        fn()
        torch.cuda.synchronize()
        import time as _t
        t0 = _t.perf_counter()
        n = 0
        while _t.perf_counter() - t0 < 0.1:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        warmup = 0
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / reps * 1e-3  # seconds per launch


def detr_geometry(device=DEV):
    shapes = torch.tensor(DETR_SHAPES, dtype=torch.int32, device=device)
    shapes._alo_shapes = list(DETR_SHAPES)   # the host copy DeformableTransformer attaches (enables the LDS-resident forward)
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]]).to(torch.int32)
    return shapes, start, int((shapes[:, 0] * shapes[:, 1]).sum())


def encoder_like_locations(N, M=8, L=4, P=4, jitter=0.5, seed=0, device=DEV):
    """Query = every pixel of every level; sampling points = the module's initial ring (head direction x (p+1) pixels)
    around the query's own position on every level, plus sub-pixel jitter.  (N, S, M, L, P, 2) float32."""
    gen = torch.Generator(device=device).manual_seed(seed)
    refs = []
    for (h, w) in DETR_SHAPES:
        ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        refs.append(torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    ref = torch.cat(refs, 0)  # (S, 2) normalised (x, y)
    S = ref.shape[0]
    ang = torch.arange(M, device=device, dtype=torch.float32) * (2 * torch.pi / M)
    ring = torch.stack([ang.cos(), ang.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]  # (M, 2)
    steps = torch.arange(1, P + 1, device=device, dtype=torch.float32)  # (P,)
    off_px = ring[:, None, None, :] * steps[None, None, :, None]  # (M,1,P,2) pixels
    norm = torch.tensor([[w, h] for h, w in DETR_SHAPES], device=device, dtype=torch.float32)  # (L,2) = (W,H)
    off = off_px / norm[None, :, None, :]  # (M,L,P,2)
    loc = ref[None, :, None, None, None, :] + off[None, None]
    loc = loc.expand(N, S, M, L, P, 2).contiguous()
    loc += (torch.rand(loc.shape, generator=gen, device=device) - 0.5) * 2 * jitter / norm[None, None, None, :, None, :]
    return loc


# "trained-like" sampling offsets — a stand-in for a trained Deformable-DETR, whose checkpoints cannot be fetched offline.  What the
# module guarantees by construction (ops/modules/ms_deform_attn.py:70-88,119-133 of the reference): offsets are a linear function
# of the query plus a bias that starts as the head-direction ring (head m looks along angle 2 pi m / 8, point p sits (p + 1) px
# out), they are measured in PIXELS OF THE LEVEL THEY SAMPLE (divided by (W_l, H_l)), so the same spread in pixels is 2^l times
# wider in the image on level l.  What training does to them (Deformable-DETR paper, fig. 5 / the released checkpoints' bias and
# weight norms): the ring survives as the mean direction, a query-dependent term of a few pixels is added, and a minority of
# points reaches far across the object.  Modelled as
#     offset_px[l, m, p] = ring[m] * (p + 1)  +  sigma_l * t3 / sqrt(3),      sigma = (1.5, 2.0, 2.5, 3.0) px on levels 0..3,
# t3 a Student-t with 3 degrees of freedom per coordinate (unit variance after the division; 0.6 % of the draws beyond 4 sigma
# against 0.006 % for a Gaussian): in image pixels the per-level spread grows 12 -> 192 px from level 0 to level 3, and one
# coordinate in 160 lands more than 4 sigma out.  Between "ring" (what a random-init model produces) and "survey" / "uniform".
TRAINED_SIGMA_PX = (1.5, 2.0, 2.5, 3.0)


def trained_like_offsets_px(N, S, gen, device=DEV, M=8, L=4, P=4):
    """(N, S, M, L, P, 2) float32 offsets in level pixels, see the comment above."""
    ang = torch.arange(M, device=device, dtype=torch.float32) * (2 * torch.pi / M)
    ring = torch.stack([ang.cos(), ang.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    steps = torch.arange(1, P + 1, device=device, dtype=torch.float32)
    mean = (ring[:, None, None, :] * steps[None, None, :, None]).expand(M, L, P, 2)
    z = torch.randn(N, S, M, L, P, 2, generator=gen, device=device)
    chi = torch.randn(N, S, M, L, P, 2, 3, generator=gen, device=device).square().sum(-1)
    t3 = z / (chi / 3.0).sqrt() / 3.0 ** 0.5
    sigma = torch.tensor(TRAINED_SIGMA_PX, device=device, dtype=torch.float32)[None, None, None, :, None, None]
    return mean[None, None] + sigma * t3


def pyramid_refs(device=DEV):
    """(S, 2) normalised (x, y) centre of every pixel of every level, pyramid order."""
    refs = []
    for (h, w) in DETR_SHAPES:
        ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        refs.append(torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    return torch.cat(refs, 0)


def msda_inputs(N, Lq, kind, dtype, seed=0):
    shapes, start, S = detr_geometry()
    gen = torch.Generator(device=DEV).manual_seed(seed)
    value = torch.randn(N, S, 8, 32, generator=gen, device=DEV).to(dtype)
    if kind == "encoder":
        assert Lq == S
        loc = encoder_like_locations(N, seed=seed)
    elif kind == "survey":   # SURVEY 8(d)'s own micro-bench inputs: loc = own pixel centre + U(-0.05, 0.05) in normalised units
        assert Lq == S
        ref = torch.cat([torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1)
                         for (h, w) in DETR_SHAPES
                         for ys, xs in [torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing="ij")]], 0)
        loc = ref[None, :, None, None, None, :] + (torch.rand(N, S, 8, 4, 4, 2, generator=gen, device=DEV) - 0.5) * 0.1
    elif kind == "trained":   # see TRAINED_SIGMA_PX above
        assert Lq == S
        wh = torch.tensor([[w, h] for h, w in DETR_SHAPES], device=DEV, dtype=torch.float32)[None, None, None, :, None, :]
        loc = pyramid_refs()[None, :, None, None, None, :] + trained_like_offsets_px(N, S, gen) / wh
    else:
        loc = torch.rand(N, Lq, 8, 4, 4, 2, generator=gen, device=DEV)
    attn = torch.softmax(torch.randn(N, Lq, 8, 16, generator=gen, device=DEV), -1).view(N, Lq, 8, 4, 4)
    return value, shapes, start, loc, attn


def msda_fwd_bytes(N, S, Lq, elem, M=8, D=32, L=4, P=4, loc_elem=4):
    return elem * (N * S * M * D + N * Lq * M * D) + loc_elem * (N * Lq * M * L * P * 3)


def msda_bwd_bytes(N, S, Lq, elem, M=8, D=32, L=4, P=4, loc_elem=4):
    return elem * (2 * N * S * M * D + N * Lq * M * D) + loc_elem * (N * Lq * M * L * P * 3 * 2)


def bench_msda_fwd(N, Lq, kind, dtype, reps):
    value, shapes, start, loc, attn = msda_inputs(N, Lq, kind, dtype)
    t = time_launches(lambda: alo_hip.msda_forward(value, shapes, start, loc, attn), reps)
    nbytes = msda_fwd_bytes(N, value.shape[1], Lq, value.element_size())
    return dict(kernel=f"msda_fwd[{kind}]", N=N, Lq=Lq, dtype=str(dtype).split(".")[-1], ms=t * 1e3,
                alg_bytes=nbytes, GBps=nbytes / t / 1e9)


def fused_inputs(N, dtype, seed=0, kind="ring"):
    """Raw module tensors for the fused-prologue entry point, encoder-like (reference points = every pixel's own centre on every
    level).  ``kind`` = where the samples fall: "trained" = TRAINED_SIGMA_PX above; "ring" = the module's initial offsets (head direction x 1..4 px, + sub-pixel
    jitter: what bench.py's random-init model produces); "survey" = SURVEY 8(d)'s micro-benchmark inputs, loc = own centre +
    U(-0.05, 0.05) of the map (+-8 x +-5 px on level 0); "uniform" = loc ~ U(0, 1) over the whole map (worst case)."""
    shapes, start, S = detr_geometry()
    gen = torch.Generator(device=DEV).manual_seed(seed)
    value = torch.randn(N, S, 8, 32, generator=gen, device=DEV).to(dtype)
    refs = []
    for (h, w) in DETR_SHAPES:
        ys, xs = torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing="ij")
        refs.append(torch.stack([(xs.reshape(-1) + 0.5) / w, (ys.reshape(-1) + 0.5) / h], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, 4, 2).contiguous()
    ang = torch.arange(8, device=DEV, dtype=torch.float32) * (2 * torch.pi / 8)
    ring = torch.stack([ang.cos(), ang.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    steps = torch.arange(1, 5, device=DEV, dtype=torch.float32)
    off = (ring[:, None, None, :] * steps[None, None, :, None]).expand(8, 4, 4, 2)
    if kind == "ring":
        offsets = off[None, None].expand(N, S, 8, 4, 4, 2) + (torch.rand(N, S, 8, 4, 4, 2, generator=gen, device=DEV) - 0.5)
    elif kind == "trained":
        offsets = trained_like_offsets_px(N, S, gen)
    else:
        wh = torch.tensor([[w, h] for h, w in DETR_SHAPES], device=DEV, dtype=torch.float32)[None, None, None, :, None, :]
        span = 0.1 if kind == "survey" else 1.0
        offsets = (torch.rand(N, S, 8, 4, 4, 2, generator=gen, device=DEV) - 0.5) * span * wh     # pixels: loc = ref + offset / (W, H)
        if kind == "uniform":
            ref = torch.full_like(ref, 0.5)
    logits = torch.randn(N, S, 8, 16, generator=gen, device=DEV)
    return value, shapes, start, offsets.to(dtype).contiguous(), logits.to(dtype), ref


def bench_msda_fused(N, dtype, reps):
    value, shapes, start, offsets, logits, ref = fused_inputs(N, dtype)
    S = value.shape[1]
    t = time_launches(lambda: alo_hip.msda_forward_fused(value, shapes, start, offsets, logits, ref), reps)
    e = value.element_size()
    nbytes = e * (N * S * 256 * 2 + N * S * 8 * 16 * 3) + ref.numel() * 4
    return dict(kernel="msda_fwd_fused[encoder]", N=N, Lq=S, dtype=str(dtype).split(".")[-1], ms=t * 1e3,
                alg_bytes=nbytes, GBps=nbytes / t / 1e9)


def bench_msda_fused_hm(N, reps, resident=True, kind="ring"):
    """Head-major path: the re-layout (+ padding mask) pass and the gather, separately."""
    value, shapes, start, offsets, logits, ref = fused_inputs(N, torch.bfloat16, kind=kind)
    S = value.shape[1]
    mask = torch.zeros(N, S, dtype=torch.bool, device=DEV)
    t0 = time_launches(lambda: alo_hip.value_head_major(value, mask), reps)
    vhm = alo_hip.value_head_major(value, mask)
    t = time_launches(lambda: alo_hip.msda_forward_fused_hm(vhm, shapes, start, offsets, logits, ref, resident=resident), reps)
    nbytes = 2 * (N * S * 256 * 2 + N * S * 8 * 16 * 3) + ref.numel() * 4
    return [dict(kernel="value_head_major[+mask]", N=N, dtype="bfloat16", ms=t0 * 1e3, alg_bytes=4 * value.numel(),
                 GBps=4 * value.numel() / t0 / 1e9),
            dict(kernel=f"msda_fwd_fused_hm[{'encoder' if kind == 'ring' else kind}]" + ("" if resident else "[plain]"), N=N, Lq=S, dtype="bfloat16", ms=t * 1e3, alg_bytes=nbytes,
                 GBps=nbytes / t / 1e9)]


def bench_msda_bwd(N, Lq, kind, dtype, reps):
    value, shapes, start, loc, attn = msda_inputs(N, Lq, kind, dtype)
    go = torch.randn(N, Lq, 256, device=DEV).to(dtype)
    t = time_launches(lambda: alo_hip.msda_backward(value, shapes, start, loc, attn, go), reps)
    nbytes = msda_bwd_bytes(N, value.shape[1], Lq, value.element_size())
    return dict(kernel=f"msda_bwd[{kind}]", N=N, Lq=Lq, dtype=str(dtype).split(".")[-1], ms=t * 1e3,
                alg_bytes=nbytes, GBps=nbytes / t / 1e9)


def corr_inputs(B, C=256, H=90, W=160, seed=0):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(B, C, H, W, generator=gen, device=DEV), torch.randn(B, C, H, W, generator=gen, device=DEV))


def bench_corr_build(B, reps, H=90, W=160, C=256):
    f1, f2 = corr_inputs(B, C, H, W)
    shapes = alo_hip.corr_level_shapes(H, W, 4)
    HW = H * W
    levels = [torch.empty((B * HW, 1, h, w), device=DEV) for h, w in shapes]
    import ctypes
    nbytes_ws = alo_hip.lib().alo_corr_build_workspace_bytes(B, C, H, W, 4)
    ws = torch.empty(nbytes_ws // 4, device=DEV)
    ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in levels])

    def run():
        rc = alo_hip.lib().alo_corr_build(f1.data_ptr(), f2.data_ptr(), ptrs, ws.data_ptr(), nbytes_ws, B, C, H, W, 4,
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, alo_hip.lib().alo_last_error()

    t = time_launches(run, reps)
    ncols = sum(h * w for h, w in shapes)
    flops_l0 = 2.0 * B * HW * HW * C  # 2*HW^2*C per pair (level 0, what the reference's matmul does)
    n3 = sum(h * w for h, w in shapes[3:])
    flops_all = 3 * 2.0 * B * HW * (HW + n3) * C  # executed on the fp16 pipe: three split products; levels >= 3 as extra columns
    nbytes = 4 * (2 * B * C * HW + B * HW * ncols)
    return dict(kernel="corr_build", B=B, ms=t * 1e3, alg_flops=flops_l0, exec_flops=flops_all,
                TFLOPs_alg=flops_l0 / t / 1e12, TFLOPs_exec=flops_all / t / 1e12, alg_bytes=nbytes,
                GBps=nbytes / t / 1e9)


def bench_corr_lookup(B, reps, H=90, W=160, C=256):
    f1, f2 = corr_inputs(B, C, H, W)
    levels = alo_hip.corr_build(f1, f2, 4)
    ys, xs = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    coords = torch.stack([xs, ys]).float()[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W, device=DEV) * 4.0
    t = time_launches(lambda: alo_hip.corr_lookup(levels, coords, 4), reps)
    HW = H * W
    nbytes = 4 * B * (HW * 324 + HW * 4 * 100 + 2 * HW)
    return dict(kernel="corr_lookup", B=B, ms=t * 1e3, alg_bytes=nbytes, GBps=nbytes / t / 1e9)


def bench_corr_lookup_bwd(B, reps, H=90, W=160):
    """The lookup's adjoint: read the 324 output gradients, read-modify-write the 4 x 10 x 10 footprints (the gradient maps)."""
    shapes = alo_hip.corr_level_shapes(H, W, 4)
    HW = H * W
    grads = [torch.zeros((B * HW, 1, h, w), device=DEV) for h, w in shapes]
    ys, xs = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    coords = torch.stack([xs, ys]).float()[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W, device=DEV) * 4.0
    gout = torch.randn(B, 324, H, W, device=DEV)
    t = time_launches(lambda: alo_hip.corr_lookup_backward(grads, coords, gout, 4), reps)
    nbytes = 4 * B * (HW * 324 + 2 * HW * 4 * 100 + 2 * HW)
    return dict(kernel="corr_lookup_backward", B=B, ms=t * 1e3, alg_bytes=nbytes, GBps=nbytes / t / 1e9)


def bench_epilogues(N, dtype, reps):
    """alo_add_layernorm on the encoder's (N*S, 256) rows and alo_bias_act on the layer1 NHWC map, next to the stock ops."""
    S = sum(h * w for h, w in DETR_SHAPES)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(N, S, 256, device=DEV, generator=g).to(dtype)
    r = torch.randn(N, S, 256, device=DEV, generator=g).to(dtype)
    pos = torch.randn(N, S, 256, device=DEV, generator=g).to(dtype)
    w, b = torch.ones(256, device=DEV, dtype=dtype), torch.zeros(256, device=DEV, dtype=dtype)
    e = x.element_size()
    out = []
    t = time_launches(lambda: alo_hip.add_layernorm(x, r, w, b, 1e-5), reps)
    out.append({"kernel": "add_layernorm[encoder rows]", "dtype": str(dtype).split(".")[-1], "ms": t * 1e3,
                "alg_bytes": 3 * x.numel() * e, "GBps": 3 * x.numel() * e / t / 1e9})
    t = time_launches(lambda: alo_hip.add_layernorm(x, r, w, b, 1e-5, pos=pos), reps)
    out.append({"kernel": "add_layernorm+pos[encoder rows]", "dtype": str(dtype).split(".")[-1], "ms": t * 1e3,
                "alg_bytes": 5 * x.numel() * e, "GBps": 5 * x.numel() * e / t / 1e9})
    t = time_launches(lambda: torch.nn.functional.layer_norm(x + r, (256,), w, b, 1e-5), reps)
    out.append({"kernel": "stock add + layer_norm", "dtype": str(dtype).split(".")[-1], "ms": t * 1e3})
    a = torch.randn(N, 256, 200, 334, device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    idn = torch.randn(N, 256, 200, 334, device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(256, device=DEV, generator=g).to(dtype)
    t = time_launches(lambda: alo_hip.bias_act_(a, bias, idn, True), reps)
    out.append({"kernel": "bias_act[+identity, relu; layer1 map]", "dtype": str(dtype).split(".")[-1], "ms": t * 1e3,
                "alg_bytes": 3 * a.numel() * e, "GBps": 3 * a.numel() * e / t / 1e9})
    t = time_launches(lambda: torch.relu_(a.add_(bias.view(1, -1, 1, 1)) + idn), reps)
    out.append({"kernel": "stock bias add_ + add + relu_", "dtype": str(dtype).split(".")[-1], "ms": t * 1e3})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="msda_enc,msda_dec,msda_rand,msda_bwd,corr_build,corr_lookup")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f32,bf16")
    ap.add_argument("--N", type=int, default=8)
    ap.add_argument("--B", type=int, default=4)
    a = ap.parse_args()
    dts = [dict(f32=torch.float32, bf16=torch.bfloat16, f64=torch.float64)[d] for d in a.dtype.split(",")]
    S = sum(h * w for h, w in DETR_SHAPES)
    tags = {k: os.environ[k] for k in os.environ if k.startswith("ALO_")}
    for w in a.which.split(","):
        res = []
        if w == "msda_enc":
            res = [bench_msda_fwd(a.N, S, "encoder", dt, a.reps) for dt in dts]
        elif w == "msda_fused":
            res = [bench_msda_fused(a.N, dt, a.reps) for dt in dts]
        elif w == "msda_fused_hm":
            res = bench_msda_fused_hm(a.N, a.reps)
        elif w == "msda_fused_hm_plain":   # the plain head-major kernel (no LDS-resident levels), for A/B
            res = bench_msda_fused_hm(a.N, a.reps, resident=False)
        elif w in ("msda_fused_hm_survey", "msda_fused_hm_uniform", "msda_fused_hm_trained"):   # the headline kernel away from the init-time ring
            kind = w.rsplit("_", 1)[1]
            res = bench_msda_fused_hm(a.N, a.reps, kind=kind)[1:] + bench_msda_fused_hm(a.N, a.reps, resident=False, kind=kind)[1:]
        elif w == "msda_rand":
            res = [bench_msda_fwd(a.N, S, "uniform", dt, a.reps) for dt in dts]
        elif w == "msda_dec":
            res = [bench_msda_fwd(a.N, 300, "uniform", dt, a.reps) for dt in dts]
        elif w == "msda_bwd":
            res = [bench_msda_bwd(4, S, "encoder", torch.float32, max(3, a.reps // 4))]
        elif w == "msda_survey":   # forward (fp32 / bf16 values) and backward on SURVEY 8(d)'s U(-0.05, 0.05) locations
            res = [bench_msda_fwd(a.N, S, "survey", dt, a.reps) for dt in dts] + [bench_msda_bwd(4, S, "survey", torch.float32, max(3, a.reps // 4))]
        elif w == "msda_trained":   # forward + backward on the trained-like offsets (TRAINED_SIGMA_PX)
            res = [bench_msda_fwd(a.N, S, "trained", dt, a.reps) for dt in dts] + [bench_msda_bwd(4, S, "trained", torch.float32, max(3, a.reps // 4))]
        elif w == "msda_bwd_rand":
            res = [bench_msda_bwd(4, S, "uniform", torch.float32, max(3, a.reps // 4))]
        elif w == "msda_bwd_bf16":   # bf16 values / grad_out (gradients fp32): ring and trained-like
            res = [bench_msda_bwd(4, S, k, torch.bfloat16, max(3, a.reps // 4)) for k in ("encoder", "trained")]
        elif w == "msda_bwd_all":    # the four distributions, fp32 (ALO_MSDA_BWD=tiled in the environment: the 4x4 tiled kernel)
            res = [bench_msda_bwd(4, S, k, torch.float32, max(3, a.reps // 4)) for k in ("encoder", "survey", "trained", "uniform")]
        elif w == "corr_build":
            res = [bench_corr_build(a.B, max(3, a.reps // 4))]
        elif w == "corr_lookup_bwd":
            res = [bench_corr_lookup_bwd(4, a.reps)]
        elif w == "corr_lookup":
            res = [bench_corr_lookup(a.B, a.reps)]
        elif w == "epilogues":
            res = [r for dt in dts for r in bench_epilogues(a.N, dt, a.reps)]
        for r in res:
            r.update(tags)
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
