"""The arithmetic behind the correlation build, restated in numpy (CPU): a per-item power-of-two scaling, a two-term fp16 split of both
operands (round to nearest) and the three cross products hi*hi + hi*lo + lo*hi accumulated in fp32 reproduce the fp32 contraction to
fp32-class accuracy at any input magnitude (aloception-oss_amd/csrc/corr.hip: corr_absmax_kernel / corr_split_kernel /
corr_gemm3_kernel; reference semantics alonet/raft/corr.py:52-60)."""
import numpy as np
import pytest


def split_exponent(absmax):
    """2^-k * absmax lands in [2^14, 2^15); 0 for an all-zero or non-finite item (split_exponent in corr.hip)."""
    if not np.isfinite(absmax) or absmax == 0:
        return 0
    e = int(np.floor(np.log2(float(absmax))))
    if e < -126:   # fp32 subnormal: exponent field 0
        return 0
    return int(np.clip(e - 14, -110, 110))


def split_fp16(x32):
    k = split_exponent(np.abs(x32).max())
    xs = np.ldexp(x32, -k).astype(np.float32)            # exact
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)  # the difference is exact in fp32, then rounded
    return hi, lo, k


def corr_split(f1, f2):
    """(C, n) x (C, m) -> (n, m) through the split path, fp32 accumulation."""
    a_hi, a_lo, ka = split_fp16(f1)
    b_hi, b_lo, kb = split_fp16(f2)
    A_hi, A_lo, B_hi, B_lo = (t.astype(np.float32) for t in (a_hi, a_lo, b_hi, b_lo))
    acc = (A_lo.T @ B_hi + A_hi.T @ B_lo) + A_hi.T @ B_hi      # small terms first, fp32 throughout
    kt = ka + kb
    scale = np.float32(1.0 / np.sqrt(f1.shape[0]))
    return acc * np.float32(np.ldexp(1.0, kt // 2)) * np.float32(np.ldexp(float(scale), kt - kt // 2))


@pytest.mark.parametrize("sa,sb", [(1.0, 1.0), (1e-3, 1e-3), (3e4, 7e5), (1e-20, 1e-12), (1e12, 1e-15), (5e18, 3e17)])
def test_two_term_fp16_split_matches_the_fp64_contraction(sa, sb):
    rng = np.random.default_rng(5)
    C, n, m = 256, 96, 80
    f1 = (rng.standard_normal((C, n)) * sa).astype(np.float32)
    f2 = (rng.standard_normal((C, m)) * sb).astype(np.float32)
    f1[:, :7] *= 1e-6   # tiny columns next to ordinary ones
    ref = (f1.astype(np.float64).T @ f2.astype(np.float64)) / np.sqrt(C)
    got = corr_split(f1, f2).astype(np.float64)
    plain = ((f1.T @ f2) / np.float32(np.sqrt(C))).astype(np.float64)   # what an fp32 matmul gives
    scale = np.abs(ref).max()
    err_split, err_plain = np.abs(got - ref).max() / scale, np.abs(plain - ref).max() / scale
    assert err_split <= 2e-6, err_split                    # fp32-class: 24-bit products, fp32 accumulation of 256 of them
    assert err_split <= 4 * err_plain + 3e-7               # and no worse than a plain fp32 matmul, up to a small factor


def test_split_terms_carry_22_bits_and_never_overflow():
    rng = np.random.default_rng(6)
    for mag in (1e-30, 1e-8, 1.0, 6.5e4, 1e9, 1e30):
        x = (rng.standard_normal(4096) * mag).astype(np.float32)
        hi, lo, k = split_fp16(x)
        assert np.isfinite(hi.astype(np.float32)).all() and np.isfinite(lo.astype(np.float32)).all()
        back = np.ldexp(hi.astype(np.float64) + lo.astype(np.float64), k)
        assert np.abs(back - x.astype(np.float64)).max() <= 2.0 ** -21 * np.abs(x).max()
        big = np.abs(x) > 2.0 ** -8 * np.abs(x).max()      # entries whose low term is a normal fp16 number: full relative accuracy
        assert (np.abs(back - x.astype(np.float64))[big] <= 2.0 ** -22 * np.abs(x.astype(np.float64))[big]).all()


def test_all_zero_and_non_finite_items_keep_exponent_zero():
    assert split_exponent(0.0) == 0 and split_exponent(np.inf) == 0 and split_exponent(np.nan) == 0
    assert split_exponent(1.0) == -14 and split_exponent(40000.0) == 1 and split_exponent(2.0 ** 14) == 0
