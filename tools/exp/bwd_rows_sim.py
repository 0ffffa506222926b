"""How many 128-byte grad_value rows reach the L2 atomic units per encoder-size backward call, as a function of the query tile a
flush covers (CPU, numpy; no GPU needed).  The tiled backward (csrc/msda.hip, msda_bwd_tiled_kernel) forms the sums of a TILE of
queries on chip and issues ONE atomic row per (tile, head, touched pixel); the chip retires 10.5 G such rows per second whatever
the pattern (tools/micro/atomic_scope.hip), so rows / 10.5 G is a floor under the kernel.  This script counts the rows exactly for
the sampling distributions of tools/kbench.py on the 1333 x 800 pyramid, N = 4, for square tiles of T x T queries of every level
("same"), and for tiles whose size follows the TARGET level (T_l x T_l level-0-equivalent pixels: "per-level").

    python tools/exp/bwd_rows_sim.py  > profiles/r05_bwd_rows_sim.txt
"""
import numpy as np

SH = [(100, 167), (50, 84), (25, 42), (13, 21)]
M, P, N = 8, 4, 4
SIGMA = (1.5, 2.0, 2.5, 3.0)          # tools/kbench.py TRAINED_SIGMA_PX
RATE = 10.5e9                         # rows / s (profiles/r04_atomic_micro.txt)


def refs():
    out = []
    for h, w in SH:
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        out.append(np.stack([(xs.ravel() + .5) / w, (ys.ravel() + .5) / h], -1))
    return out


def locations(kind, rng):
    ang = np.arange(M) * (2 * np.pi / M)
    ring = np.stack([np.cos(ang), np.sin(ang)], -1)
    ring /= np.abs(ring).max(-1, keepdims=True)
    steps = np.arange(1, P + 1)
    norm = np.array([[w, h] for h, w in SH], float)
    res = []
    for r in refs():
        Sq = r.shape[0]
        mean = ring[:, None, None, :] * steps[None, None, :, None]            # (M,1,P,2) px
        if kind == "ring":
            off = mean[None] + (rng.random((Sq, M, 4, P, 2)) - .5)
            loc = r[:, None, None, None, :] + off / norm[None, None, :, None, :]
        elif kind == "trained":
            t3 = rng.standard_t(3, (Sq, M, 4, P, 2)) / np.sqrt(3.0)
            off = mean[None] + np.array(SIGMA)[None, None, :, None, None] * t3
            loc = r[:, None, None, None, :] + off / norm[None, None, :, None, :]
        elif kind == "survey":
            loc = r[:, None, None, None, :] + (rng.random((Sq, M, 4, P, 2)) - .5) * 0.1
        else:
            loc = rng.random((Sq, M, 4, P, 2))
        res.append(loc)
    return res


def rows_per_query_head(kind, tile_of):
    """tile_of(lq, l) -> (th, tw): tile of level-lq queries whose sums for target level l are flushed together."""
    rng = np.random.default_rng(0)
    tot = np.zeros(4)
    for lq, loc in enumerate(locations(kind, rng)):
        hq, wq = SH[lq]
        qy, qx = np.divmod(np.arange(hq * wq), wq)
        for l, (H, W) in enumerate(SH):
            th, tw = tile_of(lq, l)
            tid = (qy // th) * 4096 + (qx // tw)
            x = loc[:, :, l, :, 0] * W - .5
            y = loc[:, :, l, :, 1] * H - .5
            valid = (x > -1) & (y > -1) & (x < W) & (y < H)
            x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
            keys = []
            for dy in (0, 1):
                for dx in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    ok = valid & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                    key = (tid[:, None, None] * M + np.arange(M)[None, :, None]) * (H * W) + yy * W + xx
                    keys.append(key[ok])
            tot[l] += np.unique(np.concatenate(keys)).size
    return tot / (sum(h * w for h, w in SH) * M)


def main():
    S = sum(h * w for h, w in SH)
    print(f"pyramid {SH}, S = {S}, M = {M}, N = {N}: {N * S * M / 1e6:.3f} M (query, head) pairs, {N * S * M * 64 / 1e6:.1f} M corner contributions")
    print(f"atomic rate {RATE / 1e9:.1f} G rows/s; algorithmic grad_value rows = N S M = {N * S * M / 1e6:.3f} M (one per pixel and head)\n")
    same = lambda T: (lambda lq, l: (T, T))                                   # noqa: E731
    per_level = lambda base: (lambda lq, l: (max(base[l] >> lq, 1),) * 2)     # noqa: E731
    cases = [("4 x 4 of every level (the kernel)", same(4)), ("8 x 8", same(8)), ("16 x 16", same(16)), ("32 x 32", same(32)), ("64 x 64", same(64)),
             ("per target level 8/16/32/64 px of level 0", per_level([8, 16, 32, 64])),
             ("per target level 16/32/64/128", per_level([16, 32, 64, 128]))]
    for kind in ("ring", "trained", "survey", "uniform"):
        print(f"== {kind} ==")
        for name, f in cases:
            r = rows_per_query_head(kind, f)
            rows = r.sum() * S * M * N
            print(f"  {name:45s} rows/(query, head) per target level {np.round(r, 2)}  total {r.sum():6.2f}  -> {rows / 1e6:6.2f} M rows "
                  f"= {rows / RATE * 1e3:5.2f} ms of atomics, {rows * 128 / 1e6:7.1f} MB written through ({rows * 128 / (N * S * M * 128):.1f} x grad_value)")
        print()


if __name__ == "__main__":
    main()
