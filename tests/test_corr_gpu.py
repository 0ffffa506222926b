"""Parity of the HIP correlation pyramid + lookup (through the C ABI) with the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import alo_hip
import oracle as O
from alonet.raft.corr import CorrBlock
from alonet.raft.utils.utils import coords_grid

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_g6_pyramid_and_lookup_against_reference_outputs(golden):
    g = golden("g6_corr.npz")
    blk = CorrBlock(dev(g["f1"]), dev(g["f2"]), radius=4)
    assert len(blk.corr_pyramid) == 4
    for lvl in range(4):
        got = blk.corr_pyramid[lvl].cpu().numpy()
        assert got.shape == g[f"lvl{lvl}"].shape
        np.testing.assert_allclose(got, g[f"lvl{lvl}"], rtol=0, atol=2e-5)  # fp32 summation-order noise, values O(1)
    for k in "abc":
        out = blk(dev(g["coords_" + k]))
        assert out.dtype == torch.float32 and out.is_contiguous()
        np.testing.assert_allclose(out.cpu().numpy(), g["out_" + k], rtol=0, atol=3e-5)
    vol = CorrBlock.corr(dev(g["f1"]), dev(g["f2"]))
    assert vol.shape == (1, 16, 20, 1, 16, 20)
    np.testing.assert_allclose(vol.reshape(320, 1, 16, 20).cpu().numpy(), g["lvl0"], rtol=0, atol=2e-5)


def test_g6_odd_sizes_batched_radius3(golden):
    g = golden("g6_corr.npz")
    blk = CorrBlock(dev(g["f1o"]), dev(g["f2o"]), radius=3)
    for lvl in range(4):
        np.testing.assert_allclose(blk.corr_pyramid[lvl].cpu().numpy(), g[f"lvl{lvl}o"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(blk(dev(g["coords_o"])).cpu().numpy(), g["out_o"], rtol=0, atol=1e-5)


def test_lookup_alone_on_the_reference_pyramid(golden):
    """Feed the reference's own pyramid to the lookup kernel: isolates the gather from the GEMM."""
    g = golden("g6_corr.npz")
    levels = [dev(g[f"lvl{lvl}"]) for lvl in range(4)]
    for k in "abc":
        out = alo_hip.corr_lookup(levels, dev(g["coords_" + k]), 4).cpu().numpy()
        np.testing.assert_allclose(out, g["out_" + k], rtol=0, atol=5e-6)  # separable lerp vs 4-tap sum, |v| <= 4


@pytest.mark.parametrize("B,C,H,W,r,L", [(2, 256, 24, 32, 4, 4), (1, 64, 19, 23, 4, 3), (3, 37, 16, 18, 2, 4),
                                         (1, 128, 33, 47, 1, 2), (1, 16, 40, 40, 7, 4), (2, 8, 16, 16, 0, 1),
                                         (3, 3, 5, 7, 1, 2)])   # last: C*H*W odd, so batch items 1.. are not 16-byte aligned
def test_build_and_lookup_vs_oracle(B, C, H, W, r, L):
    rng = np.random.default_rng(B * 1000 + C + H)
    f1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    ref_pyr = O.corr_pyramid(f1, f2, L)
    levels = alo_hip.corr_build(dev(f1), dev(f2), L)
    for lvl in range(L):
        got = levels[lvl].cpu().numpy()
        assert got.shape == ref_pyr[lvl].shape
        np.testing.assert_allclose(got, ref_pyr[lvl], rtol=0, atol=3e-5)
    grid = coords_grid(B, H, W).numpy()
    for spread in (0.0, 3.0, 60.0):
        coords = (grid + rng.standard_normal(grid.shape) * spread).astype(np.float32)
        out = alo_hip.corr_lookup(levels, dev(coords), r).cpu().numpy()
        ref = O.corr_lookup([lv.cpu().numpy() for lv in levels], coords, r)  # same pyramid: isolates the gather
        assert out.shape == (B, L * (2 * r + 1) ** 2, H, W)
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5)


def test_non_finite_and_huge_coordinates_read_as_zero():
    rng = np.random.default_rng(9)
    f = rng.standard_normal((1, 32, 16, 16)).astype(np.float32)
    levels = alo_hip.corr_build(dev(f), dev(f), 4)
    coords = coords_grid(1, 16, 16).numpy().copy()
    coords[0, 0, 0, 0] = np.nan
    coords[0, 1, 0, 1] = np.inf
    coords[0, 0, 0, 2] = 3e9
    coords[0, 1, 0, 3] = -3e9
    out = alo_hip.corr_lookup(levels, dev(coords), 4).cpu().numpy()
    assert np.all(out[0, :, 0, :4] == 0) and np.isfinite(out).all()
    ref = O.corr_lookup([lv.cpu().numpy() for lv in levels], coords, 4)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5)


def test_full_size_properties_720p():
    """BASELINE config 3 grid (90x160, C=256), one pair: properties that need no CPU pass over 830 MB."""
    B, C, H, W = 1, 256, 90, 160
    gen = torch.Generator(device="cpu").manual_seed(5)
    f1 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    blk = CorrBlock(f1, f2)
    assert [tuple(p.shape) for p in blk.corr_pyramid] == [(14400, 1, 90, 160), (14400, 1, 45, 80),
                                                          (14400, 1, 22, 40), (14400, 1, 11, 20)]
    lvl0 = blk.corr_pyramid[0].view(H * W, H * W)
    # (1) transpose symmetry: corr(f1, f2)[i, j] == corr(f2, f1)[j, i]: the same six split products per channel, the two mixed
    # pairs accumulated in swapped order -> equal to fp32 rounding of the accumulation, not to the bit
    swapped = CorrBlock.corr(f2, f1).view(H * W, H * W)
    assert (lvl0 - swapped.t()).abs().max().item() <= 2e-6
    # (2) a slab of rows against torch's own fp32 matmul
    rows = torch.arange(0, H * W, 97, device=DEV)
    ref = (f1.view(C, -1)[:, rows].t().double() @ f2.view(C, -1).double()) / 16.0
    assert (lvl0[rows].double() - ref).abs().max().item() <= 2e-5
    # (3) pyramid consistency: level l+1 equals the 2x2 mean of level l (pool-after-correlate) to fp32 rounding
    for lvl in range(3):
        pooled = torch.nn.functional.avg_pool2d(blk.corr_pyramid[lvl][::53], 2, stride=2)
        assert (blk.corr_pyramid[lvl + 1][::53] - pooled).abs().max().item() <= 2e-5
    # (4) lookup at integer coordinates: centre tap of level 0 is the volume entry itself, window taps its neighbours
    coords = coords_grid(B, H, W, device=DEV)
    out = blk(coords)
    assert out.shape == (1, 324, 90, 160)
    centre = out[0, 4 * 9 + 4].reshape(-1)
    diag = lvl0.diagonal()
    assert (centre - diag).abs().max().item() <= 1e-4 * diag.abs().max().item()
    right = out[0, 5 * 9 + 4, :, :-1].reshape(-1)  # a = 5 -> x + 1
    ii = torch.arange(H * W, device=DEV).view(H, W)[:, :-1].reshape(-1)
    assert (right - lvl0[ii, ii + 1]).abs().max().item() <= 1e-4 * diag.abs().max().item()
    down = out[0, 4 * 9 + 5, :-1, :].reshape(-1)  # c = 5 -> y + 1
    jj = torch.arange(H * W, device=DEV).view(H, W)[:-1, :].reshape(-1)
    assert (down - lvl0[jj, jj + W]).abs().max().item() <= 1e-4 * diag.abs().max().item()
    # (5) sub-pixel lookups against the oracle on a strided subset of queries (oracle on 1/64 of the volume)
    coords2 = coords + torch.randn(B, 2, H, W, generator=gen).to(DEV) * 3.0
    out2 = blk(coords2).cpu().numpy()
    sel = np.arange(0, H * W, 64)
    pyr_sel = [p[sel].cpu().numpy() for p in blk.corr_pyramid]
    csel = coords2.view(2, -1)[:, sel].cpu().numpy()
    # oracle expects (B,2,H,W) coords with one volume row per query: present the subset as a 1 x len(sel) grid
    # whose volumes keep their true (h_l, w_l) shapes
    import ctypes
    n = len(sel)
    o = np.empty((1, 324, 1, n), np.float32)
    ptrs = (ctypes.c_void_p * 4)(*[p.ctypes.data for p in pyr_sel])
    hw = np.array([[p.shape[2], p.shape[3]] for p in pyr_sel], np.int32)
    cc = np.ascontiguousarray(csel.reshape(1, 2, 1, n))
    O.lib().oracle_corr_lookup(ptrs, hw.ctypes.data_as(ctypes.c_void_p), cc.ctypes.data_as(ctypes.c_void_p),
                               o.ctypes.data_as(ctypes.c_void_p), 1, 1, n, 4, 4)
    np.testing.assert_allclose(out2.reshape(324, -1)[:, sel], o.reshape(324, n), rtol=0, atol=2e-5)


@pytest.mark.parametrize("sa,sb", [(1e-3, 1e-3), (3e4, 7e5), (1e-20, 1e-12), (1e12, 1e-15), (5e18, 3e17)])
def test_build_is_accurate_at_any_magnitude(sa, sb):
    """The operands travel as two fp16 terms after a power-of-two scaling (per pixel since round 5): the volume must keep fp32-class relative accuracy
    from 1e-20 to 1e+18, with batch items of very different magnitude side by side, small entries next to large ones, and an
    all-zero item."""
    rng = np.random.default_rng(77)
    B, C, H, W = 3, 96, 12, 20
    f1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    f1[0] *= sa; f2[0] *= sb
    f1[1] *= sa * 1e-3; f2[1] *= sb * 4e2          # another magnitude in the same launch
    f1[1, :, :3] *= 1e-6                           # tiny rows next to ordinary ones: they only need accuracy relative to the item
    f1[2] = 0.0                                    # an all-zero item
    a64, b64 = f1.astype(np.float64).reshape(B, C, -1), f2.astype(np.float64).reshape(B, C, -1)
    vol = (np.einsum("bci,bcj->bij", a64, b64) / np.sqrt(C)).reshape(B * H * W, 1, H, W)   # corr.py:52-60 in float64
    ref = [vol]
    for _ in range(3):                                                                        # corr.py:20-27
        p = ref[-1]
        h, w = p.shape[2] // 2, p.shape[3] // 2
        ref.append(p[:, :, :2 * h, :2 * w].reshape(-1, 1, h, 2, w, 2).mean(axis=(3, 5)))
    levels = alo_hip.corr_build(dev(f1), dev(f2), 4)
    n = H * W
    for lvl in range(4):
        got = levels[lvl].cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all()
        for b in range(B):
            r = ref[lvl][b * n:(b + 1) * n]
            scale = max(np.abs(r).max(), 1e-300)
            if b == 2:
                assert not got[b * n:(b + 1) * n].any()
            else:   # 3e-6 of the item's largest entry: fp32 accumulation of 96 products, nothing worse
                assert np.abs(got[b * n:(b + 1) * n] - r).max() <= 3e-6 * scale, (lvl, b)


def test_build_accuracy_is_relative_to_each_row_and_column_not_to_the_item():
    """Round-4 verdict ("fp32 only relative to the item's largest entry"): every pixel's feature vector is now scaled by its OWN power
    of two, so an entry of the volume is accurate relative to |f1_i| |f2_j| — what torch.matmul's fp32 dot product gives (its
    error is relative to sum_c |a_c b_c|) — however dim pixel i or j is next to the item's brightest.  Pixels span 24 orders of
    magnitude inside ONE item; bound: 4e-6 of sum_c |f1_ci f2_cj| / sqrt(C) per entry (fp32 accumulation of 96 products of 22-bit
    operands), levels 1-3 against the pooled float64 volume with the pooled bound."""
    rng = np.random.default_rng(79)
    B, C, H, W = 2, 96, 12, 20
    f1 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    f2 = rng.standard_normal((B, C, H, W)).astype(np.float32)
    f1 *= (10.0 ** rng.uniform(-12, 12, (B, 1, H, W))).astype(np.float32)      # every pixel its own magnitude
    f2 *= (10.0 ** rng.uniform(-12, 12, (B, 1, H, W))).astype(np.float32)
    f2[1, :, 5, 7] = 0.0                                                         # an all-zero pixel among them
    a64, b64 = f1.astype(np.float64).reshape(B, C, -1), f2.astype(np.float64).reshape(B, C, -1)
    vol = (np.einsum("bci,bcj->bij", a64, b64) / np.sqrt(C)).reshape(B * H * W, 1, H, W)
    bound = (np.einsum("bci,bcj->bij", np.abs(a64), np.abs(b64)) / np.sqrt(C)).reshape(B * H * W, 1, H, W)
    ref, bnd = [vol], [bound]
    for _ in range(3):
        h, w = ref[-1].shape[2] // 2, ref[-1].shape[3] // 2
        pool = lambda p: p[:, :, :2 * h, :2 * w].reshape(-1, 1, h, 2, w, 2).mean(axis=(3, 5))   # noqa: E731
        ref.append(pool(ref[-1]))
        bnd.append(pool(bnd[-1]))
    levels = alo_hip.corr_build(dev(f1), dev(f2), 4)
    for lvl in range(4):
        got = levels[lvl].cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all()
        err = np.abs(got - ref[lvl])
        assert (err <= 4e-6 * bnd[lvl] + 1e-300).all(), (lvl, float((err / (bnd[lvl] + 1e-300)).max()))
    n = H * W
    assert not levels[0].cpu().numpy().reshape(B, n, n)[1, :, 5 * W + 7].any()


def test_build_propagates_non_finite_features_like_the_reference():
    rng = np.random.default_rng(78)
    f1 = rng.standard_normal((2, 32, 8, 12)).astype(np.float32)
    f2 = rng.standard_normal((2, 32, 8, 12)).astype(np.float32)
    f1[0, 3, 2, 5] = np.inf
    f2[0, 7, 1, 1] = np.nan
    levels = alo_hip.corr_build(dev(f1), dev(f2), 2)
    n = 8 * 12
    got = levels[0].cpu().numpy().reshape(2, n, n)
    assert not np.isfinite(got[0, 2 * 12 + 5]).any()      # the row of the infinite feature
    assert np.isnan(got[0, :, 1 * 12 + 1]).all()           # the column of the NaN feature
    ref1 = O.corr_pyramid(f1[1:], f2[1:], 2)[0]
    np.testing.assert_allclose(got[1].reshape(n, 1, 8, 12), ref1, rtol=0, atol=3e-5)   # the other item is untouched
    # ... and so is the REST of the spoiled item: its power-of-two scale comes from the largest FINITE magnitude, so finite
    # features above fp16's range (1e5 > 65504) sitting next to the non-finite ones neither overflow nor lose accuracy
    f1[0, 5, 6, 7] = 1.0e5
    f2[0, 9, 4, 4] = -3.0e5
    got = alo_hip.corr_build(dev(f1), dev(f2), 1)[0].cpu().numpy().reshape(2, n, n)
    a64, b64 = f1[0].astype(np.float64).reshape(32, n), f2[0].astype(np.float64).reshape(32, n)
    with np.errstate(invalid="ignore", over="ignore"):
        ref0 = a64.T @ b64 / np.sqrt(32.0)
    clean = np.isfinite(ref0)
    assert clean.sum() == (n - 1) * (n - 1)                 # everything but one row and one column
    assert np.isfinite(got[0][clean]).all()
    assert np.abs(got[0][clean] - ref0[clean]).max() <= 3e-6 * np.abs(ref0[clean]).max()
    assert not np.isfinite(got[0][~clean]).any()


def test_batch_items_are_independent():
    rng = np.random.default_rng(21)
    f1 = dev(rng.standard_normal((3, 64, 16, 24)).astype(np.float32))
    f2 = dev(rng.standard_normal((3, 64, 16, 24)).astype(np.float32))
    coords = coords_grid(3, 16, 24, device=DEV) + 1.5
    full = CorrBlock(f1, f2)
    solo = CorrBlock(f1[1:2].contiguous(), f2[1:2].contiguous())
    n = 16 * 24
    for lvl in range(4):
        assert torch.equal(full.corr_pyramid[lvl][n:2 * n], solo.corr_pyramid[lvl])
    assert torch.equal(full(coords)[1], solo(coords[1:2])[0])


# ---- BASELINE configs[2] at its own batch size ------------------------------------------------------------------------------------
def test_config3_build_and_lookup_at_batch4_720p():
    """B = 4 pairs of 256 x 90 x 160 features — where level 0 (3.3 GB) crosses 2^31 bytes and the per-item power-of-two scales,
    the flat-block -> (item, column tile, row tile) mapping and the lookup's slab offsets are all exercised together (reference:
    alonet/raft/corr.py:12-60).  Items get different magnitudes on purpose."""
    B, C, H, W = 4, 256, 90, 160
    n = H * W
    gen = torch.Generator(device="cpu").manual_seed(31)
    f1 = torch.randn(B, C, H, W, generator=gen)
    f2 = torch.randn(B, C, H, W, generator=gen)
    for b, s in enumerate((1.0, 37.0, 0.02, 5.0)):
        f1[b] *= s
        f2[b] /= s if b != 3 else 1.0
    f1, f2 = f1.to(DEV), f2.to(DEV)
    blk = CorrBlock(f1, f2)
    assert [tuple(p.shape) for p in blk.corr_pyramid] == [(B * n, 1, 90, 160), (B * n, 1, 45, 80), (B * n, 1, 22, 40), (B * n, 1, 11, 20)]
    assert blk.corr_pyramid[0].numel() * 4 > 2 ** 31
    # (1) row slabs of items 0 and 3 (first, strided middle, last rows) against a float64 product
    for b in (0, 3):
        rows = torch.cat([torch.arange(0, 64), torch.arange(64, n - 64, 211), torch.arange(n - 64, n)]).to(DEV)
        ref = (f1[b].view(C, n)[:, rows].t().double() @ f2[b].view(C, n).double()) / 16.0
        got = blk.corr_pyramid[0].view(B, n, n)[b][rows].double()
        assert (got - ref).abs().max().item() <= 3e-6 * ref.abs().max().item(), b
    # (2) levels 1-3 against pooling of the level below (pool-after-correlate), every 41st volume of every item
    for lvl in range(3):
        pooled = torch.nn.functional.avg_pool2d(blk.corr_pyramid[lvl][::41], 2, stride=2)
        scale = blk.corr_pyramid[lvl][::41].abs().max().item()
        assert (blk.corr_pyramid[lvl + 1][::41] - pooled).abs().max().item() <= 2e-6 * scale, lvl
    # (3) item 3 of the batched build is BIT-equal to a solo build of that pair, and so is its lookup
    solo = CorrBlock(f1[3:4].contiguous(), f2[3:4].contiguous())
    for lvl in range(4):
        assert torch.equal(blk.corr_pyramid[lvl][3 * n:], solo.corr_pyramid[lvl]), lvl
    coords = coords_grid(B, H, W, device=DEV) + torch.randn(B, 2, H, W, generator=gen).to(DEV) * 4.0
    out = blk(coords)
    assert out.shape == (B, 324, H, W) and torch.isfinite(out).all()
    assert torch.equal(out[3], solo(coords[3:4].contiguous())[0])
    del solo
    # (4) lookup of a strided query subset of item 3 against the C oracle fed the same volumes
    import ctypes
    sel = np.arange(0, n, 48)
    pyr_sel = [np.ascontiguousarray(p[3 * n:][sel].cpu().numpy()) for p in blk.corr_pyramid]
    cc = np.ascontiguousarray(coords[3].view(2, -1)[:, sel].cpu().numpy().reshape(1, 2, 1, len(sel)))
    o = np.empty((1, 324, 1, len(sel)), np.float32)
    ptrs = (ctypes.c_void_p * 4)(*[p.ctypes.data for p in pyr_sel])
    hw = np.array([[p.shape[2], p.shape[3]] for p in pyr_sel], np.int32)
    O.lib().oracle_corr_lookup(ptrs, hw.ctypes.data_as(ctypes.c_void_p), cc.ctypes.data_as(ctypes.c_void_p),
                               o.ctypes.data_as(ctypes.c_void_p), 1, 1, len(sel), 4, 4)
    got = out[3].reshape(324, -1)[:, torch.from_numpy(sel).to(DEV)].cpu().numpy()
    np.testing.assert_allclose(got, o.reshape(324, len(sel)), rtol=0, atol=2e-5 * max(1.0, float(np.abs(o).max())))


def test_hip_corr_block_under_autograd_has_the_gradients_of_the_torch_formulation():
    """The reference's CorrBlock is differentiable torch code (corr.py:12-60).  Here the forward is the HIP kernels whatever the
    grad mode; under autograd the lookup's backward is the kernel's adjoint (accumulated maps), the build's two GEMMs per level, the
    coordinates' the alo_corr_lookup_backward_coords kernel: gradients w.r.t. both feature maps (through the lookup AND through pyramid tensors used
    directly in the loss) and w.r.t. the coordinates must equal those of TorchCorrBlock."""
    from alonet.raft.corr import TorchCorrBlock

    gen = torch.Generator(device="cpu").manual_seed(17)
    B, C, H, W = 2, 48, 16, 24
    f1 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    coords = (coords_grid(B, H, W, device=DEV) + torch.randn(B, 2, H, W, generator=gen).to(DEV) * 2.0)
    wts = torch.randn(B, 324, H, W, generator=gen).to(DEV)
    grads = {}
    for name, cls in (("hip", CorrBlock), ("torch", TorchCorrBlock)):
        a, b, c = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True), coords.clone().requires_grad_(True)
        blk = cls(a, b, radius=4)
        out = blk(c)
        assert out.requires_grad
        loss = (out * wts).sum() + sum(p.square().sum() for p in blk.corr_pyramid) * 1e-3
        loss.backward()
        grads[name] = (out.detach(), a.grad, b.grad, c.grad)
    assert (grads["hip"][0] - grads["torch"][0]).abs().max().item() <= 5e-5      # forward: kernels vs torch ops
    for got, want in zip(grads["hip"][1:], grads["torch"][1:]):
        assert got is not None and (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())
    # RAFT's own call pattern: coords detached, only the feature maps train
    a = f1.clone().requires_grad_(True)
    CorrBlock(a, f2)(coords).sum().backward()
    assert a.grad is not None and torch.isfinite(a.grad).all()
    # no autograd graph: the plain kernels, no Function objects in the way
    with torch.no_grad():
        assert not CorrBlock(a, f2)(coords).requires_grad


@pytest.mark.parametrize("B,H,W,r,L", [(1, 16, 24, 4, 4), (2, 9, 13, 3, 2), (1, 17, 18, 1, 3), (2, 8, 8, 0, 1), (1, 33, 20, 2, 4), (1, 40, 64, 7, 3)])
def test_lookup_backward_is_the_adjoint_of_the_lookup(B, H, W, r, L):
    """alo_corr_lookup_backward against autograd through the torch formulation (the reference's bilinear_sampler = grid_sample,
    corr.py:29-50) with respect to every pyramid level: coordinates inside, on integer positions, across every border, far outside
    and non-finite; two calls accumulate; the result does not depend on the order of anything (plain loads and stores)."""
    from alonet.raft.corr import lookup_torch

    gen = torch.Generator(device="cpu").manual_seed(100 * H + W + r)
    pyr = []
    for lvl in range(L):
        h, w = H // 2 ** lvl, W // 2 ** lvl
        pyr.append(torch.randn(B * H * W, 1, h, w, generator=gen).to(DEV))
    coords = coords_grid(B, H, W, device=DEV) + torch.randn(B, 2, H, W, generator=gen).to(DEV) * 3.0
    coords[:, :, 0, :] = torch.round(coords[:, :, 0, :])                       # integer positions: a weight is exactly zero
    coords[:, 0, 1, :] = -float(r) - 2.5                                       # window entirely left of the map
    coords[:, 1, 2, :] = float(H) + 0.25                                       # window straddling the lower border
    if H > 4:
        coords[:, :, 3, 0] = float("nan")
        coords[:, :, 3, 1] = 3e7
    gout = torch.randn(B, L * (2 * r + 1) ** 2, H, W, generator=gen).to(DEV)
    leaves = [p.clone().requires_grad_(True) for p in pyr]
    ref_out = lookup_torch(leaves, coords, r)
    finite = torch.isfinite(ref_out)                                           # the torch formulation turns NaN coordinates into NaN features;
    want = torch.autograd.grad(ref_out, leaves, torch.where(finite, gout, torch.zeros_like(gout)))   # the kernels read them as zeros
    want = [torch.nan_to_num(g, nan=0.0) for g in want]
    got = [torch.zeros_like(p) for p in pyr]
    assert alo_hip.corr_lookup_backward(got, coords, gout, r) is got
    for lvl, (g, w_) in enumerate(zip(got, want)):
        assert (g - w_).abs().max().item() <= 2e-5 * max(1.0, w_.abs().max().item()), lvl
    again = [g.clone() for g in got]
    alo_hip.corr_lookup_backward(again, coords, gout, r)                       # accumulates
    for g, a in zip(got, again):
        assert torch.equal(a, 2 * g)
    # <lookup(P), G> == <P, lookup_backward(G)> for the kernels themselves (fp32 sums of ~1e5 terms: 1e-4 relative)
    fwd = alo_hip.corr_lookup(pyr, coords, r)
    lhs = (fwd.double() * gout.double()).sum().item()
    rhs = sum((p.double() * g.double()).sum().item() for p, g in zip(pyr, got))
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


@pytest.mark.parametrize("B,H,W,r,L", [(1, 16, 24, 4, 4), (2, 9, 13, 3, 2), (1, 17, 18, 1, 3), (2, 8, 8, 0, 1), (1, 40, 64, 7, 3)])
def test_lookup_backward_with_respect_to_the_coordinates(B, H, W, r, L):
    """alo_corr_lookup_backward_coords against autograd through the torch formulation (grid_sample's gradient with respect to the
    grid, chained through the reference's coordinate arithmetic): random sub-pixel positions inside, across the borders and far
    outside the maps.  Positions within 1e-3 of an integer are left out of the comparison — the interpolant has a kink there and
    fp32 arithmetic of two formulations may land on either side of it."""
    from alonet.raft.corr import lookup_torch

    gen = torch.Generator(device="cpu").manual_seed(7 * H + W + r)
    pyr = [torch.randn(B * H * W, 1, H // 2 ** lvl, W // 2 ** lvl, generator=gen).to(DEV) for lvl in range(L)]
    coords = coords_grid(B, H, W, device=DEV) + torch.randn(B, 2, H, W, generator=gen).to(DEV) * 3.0
    coords[:, 0, 1, :] = -float(r) - 2.5
    coords[:, 1, 2, :] = float(H) + 0.25
    gout = torch.randn(B, L * (2 * r + 1) ** 2, H, W, generator=gen).to(DEV)
    c = coords.clone().requires_grad_(True)
    (want,) = torch.autograd.grad(lookup_torch(pyr, c, r), c, gout)
    got = alo_hip.corr_lookup_backward_coords(pyr, coords, gout, r)
    assert got.shape == want.shape
    safe = torch.ones_like(coords, dtype=torch.bool)
    for lvl in range(L):
        p = coords / 2 ** lvl
        near = ((p - p.round()).abs() < 1e-3).any(dim=1, keepdim=True)
        safe &= ~near
    assert safe.float().mean().item() > 0.9
    err = ((got - want).abs() * safe).max().item()
    assert err <= 1e-4 * max(1.0, want.abs().max().item()), err
    # the block: coordinates attached, features detached, and both at once
    f1 = torch.randn(B, 24, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, 24, H, W, generator=gen).to(DEV)
    from alonet.raft.corr import TorchCorrBlock

    res = {}
    for name, cls in (("hip", CorrBlock), ("torch", TorchCorrBlock)):
        a, cc = f1.clone().requires_grad_(True), coords.clone().requires_grad_(True)
        out = cls(a, f2, num_levels=L, radius=r)(cc)
        ga, gc = torch.autograd.grad((out * gout).sum(), (a, cc))
        res[name] = (ga, gc)
    assert (res["hip"][0] - res["torch"][0]).abs().max().item() <= 2e-4 * max(1.0, res["torch"][0].abs().max().item())
    assert ((res["hip"][1] - res["torch"][1]).abs() * safe).max().item() <= 2e-4 * max(1.0, res["torch"][1].abs().max().item())
    cc = coords.clone().requires_grad_(True)
    out = CorrBlock(f1, f2, num_levels=L, radius=r)(cc)                      # only the coordinates: the dense-function route
    (gc,) = torch.autograd.grad((out * gout).sum(), cc)
    assert ((gc - res["torch"][1]).abs() * safe).max().item() <= 2e-4 * max(1.0, res["torch"][1].abs().max().item())


def test_corr_block_gradients_over_many_lookups_and_repeated_backward():
    """RAFT's pattern: one pyramid, many lookups at moving (detached) coordinates, one backward — the lookups accumulate into shared
    gradient maps that the build node turns into feature gradients with two GEMMs per level.  Gradients equal those of the torch
    formulation; a second backward through the retained graph gives the same result (the maps start from zero again); odd sizes."""
    from alonet.raft.corr import TorchCorrBlock

    gen = torch.Generator(device="cpu").manual_seed(5)
    B, C, H, W = 2, 40, 15, 22
    f1 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, C, H, W, generator=gen).to(DEV)
    steps = [coords_grid(B, H, W, device=DEV) + torch.randn(B, 2, H, W, generator=gen).to(DEV) * (1.0 + k) for k in range(5)]
    wts = [torch.randn(B, 3 * 49, H, W, generator=gen).to(DEV) for _ in steps]
    res = {}
    for name, cls in (("hip", CorrBlock), ("torch", TorchCorrBlock)):
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        blk = cls(a, b, num_levels=3, radius=3)
        loss = sum((blk(c) * w_).sum() for c, w_ in zip(steps, wts))
        loss.backward(retain_graph=True)
        first = (a.grad.clone(), b.grad.clone())
        a.grad = b.grad = None
        loss.backward()
        res[name] = (first, (a.grad, b.grad))
    for k in range(2):
        assert torch.equal(res["hip"][0][k], res["hip"][1][k])                 # deterministic, and nothing left over from the first pass
        want = res["torch"][0][k]
        assert (res["hip"][0][k] - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())
    # only one of the two feature maps trains
    a = f1.clone().requires_grad_(True)
    blk = CorrBlock(a, f2, num_levels=3, radius=3)
    sum((blk(c) * w_).sum() for c, w_ in zip(steps, wts)).backward()
    assert (a.grad - res["torch"][0][0]).abs().max().item() <= 2e-4 * max(1.0, res["torch"][0][0].abs().max().item())
    b = f2.clone().requires_grad_(True)
    blk = CorrBlock(f1, b, num_levels=3, radius=3)
    sum((blk(c) * w_).sum() for c, w_ in zip(steps, wts)).backward()
    assert (b.grad - res["torch"][0][1]).abs().max().item() <= 2e-4 * max(1.0, res["torch"][0][1].abs().max().item())
    # CorrBlock.corr (the volume alone) is differentiable the same way
    a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    g = torch.randn(B, H, W, 1, H, W, generator=gen).to(DEV)
    (CorrBlock.corr(a, b) * g).sum().backward()
    a2, b2 = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    (TorchCorrBlock.corr(a2, b2) * g).sum().backward()
    assert (a.grad - a2.grad).abs().max().item() <= 2e-4 * max(1.0, a2.grad.abs().max().item())
    assert (b.grad - b2.grad).abs().max().item() <= 2e-4 * max(1.0, b2.grad.abs().max().item())


def test_g18_hip_corr_block_has_the_reference_gradients(golden):
    """The HIP block's gradients (adjoint kernel + GEMMs; coordinates through the alo_corr_lookup_backward_coords kernel) against the REFERENCE's own
    CorrBlock differentiated by autograd (G18, float64 run): odd sizes, 3 levels, radius 2, three lookups, windows off the map."""
    g = golden("g18_corr_grad.npz")
    f1 = torch.from_numpy(g["f1"]).to(DEV).requires_grad_(True)
    f2 = torch.from_numpy(g["f2"]).to(DEV).requires_grad_(True)
    coords, wts = torch.from_numpy(g["coords"]).to(DEV), torch.from_numpy(g["weights"]).to(DEV)
    blk = CorrBlock(f1, f2, num_levels=3, radius=2)
    outs = [blk(c) for c in coords]
    for o, want in zip(outs, torch.from_numpy(g["out"]).to(DEV)):
        assert (o - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    g1, g2 = torch.autograd.grad(sum((o * w).sum() for o, w in zip(outs, wts)), (f1, f2))
    for got, key in ((g1, "grad_f1"), (g2, "grad_f2")):
        want = torch.from_numpy(g[key]).to(DEV)
        assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), key
    c = coords[1].clone().requires_grad_(True)
    out = CorrBlock(f1.detach(), f2.detach(), num_levels=3, radius=2)(c)
    (gc,) = torch.autograd.grad((out * wts[1]).sum(), c)
    want = torch.from_numpy(g["grad_coords1"]).to(DEV)
    # row 0 of these coordinates sits on integer positions, where the interpolant has a kink: its one-sided derivatives differ and
    # fp32 / fp64 evaluation of the position may land on either side — compared away from it
    assert (gc - want)[:, :, 1:].abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


def test_raft_is_trainable_with_the_default_corr_block():
    """`RAFT()` (default corr_block = the HIP CorrBlock) under autograd: the feature encoder receives gradients through the
    correlation volume — fine-tuning is drop-in, as with the reference's torch CorrBlock."""
    import aloscene
    from alonet.raft import RAFT

    torch.manual_seed(0)
    model = RAFT().to(DEV).train()
    model.freeze_bn()
    mk = lambda x: aloscene.Frame(x, normalization="minmax_sym", names=("B", "C", "H", "W")).to(DEV)  # noqa: E731
    f1 = torch.rand(1, 3, 128, 160) * 2 - 1
    f2 = torch.roll(f1, shifts=(2, -3), dims=(2, 3))
    outs = model(mk(f1), mk(f2), iters=2)
    loss = sum(o["up_flow"].abs().mean() for o in outs)
    loss.backward()
    g = model.fnet.conv1.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max().item() > 0


def test_partial_backward_does_not_leak_into_the_next_one():
    """Round-4 advisor finding on the real kernels: ``autograd.grad(loss, coords)`` runs the lookups' backward (which accumulates
    pyramid-gradient maps) without reaching the build node; the following full backward must not see those maps again."""
    from alonet.raft.corr import TorchCorrBlock

    gen = torch.Generator(device="cpu").manual_seed(23)
    f1 = torch.randn(1, 32, 12, 16, generator=gen).to(DEV)
    f2 = torch.randn(1, 32, 12, 16, generator=gen).to(DEV)
    cs = [coords_grid(1, 12, 16, device=DEV) + torch.randn(1, 2, 12, 16, generator=gen).to(DEV) * 2.0 for _ in range(2)]

    def grads(cls, partial_first):
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        cc = [c.clone().requires_grad_(True) for c in cs]
        blk = cls(a, b, num_levels=3, radius=2)
        loss = sum(blk(c).square().sum() for c in cc)
        if partial_first:
            torch.autograd.grad(loss, cc, retain_graph=True)
        loss.backward()
        return a.grad, b.grad

    want = grads(TorchCorrBlock, False)
    for partial_first in (False, True):
        got = grads(CorrBlock, partial_first)
        for g, w in zip(got, want):
            assert (g - w).abs().max().item() <= 2e-4 * w.abs().max().item(), partial_first
