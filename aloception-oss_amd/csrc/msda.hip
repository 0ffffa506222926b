// Multi-scale deformable attention for gfx950 (MI355X): forward gather and backward scatter.
//
// What is computed is fixed by the reference op (alonet/deformable_detr/ops/src/cuda/ms_deform_im2col_cuda.cuh:
// forward :237-299 + bilinear :33-84; backward rule :87-159, channel reductions :301-403).  How it is computed is not
// the reference's one-thread-per-output-element scheme:
//
//   * a workgroup owns a contiguous run of (query, head) "pairs" of ONE batch item.  For every pair the L*P sampling
//     points are turned into descriptors ONCE (corner offsets + corner weights already multiplied by the attention
//     weight) by all 256 threads and parked in LDS; the reference recomputes them in each of the D channel threads.
//   * the channel dimension is the coalesced one: G lanes x VEC elements = one 16-byte buffer load per lane, so a
//     corner of one head is a single 128-byte (fp32, D=32) contiguous request.
//   * corners outside the map are not branched around: their descriptor carries an out-of-range byte offset and the
//     buffer resource's bounds check returns 0 without touching memory (zero padding for free, NaN-safe).
//   * launch-order block ids are remapped so that every XCD (private L2) walks one contiguous range of the batch —
//     with N = 8 each XCD's L2 holds exactly one image's value map.
//   * backward: lanes reduce d/d(loc), d/d(attn) over channels with wave shuffles (no LDS, no block barriers, no
//     serial thread-0 sum), grad_value goes out through hardware fp32/fp64 atomics.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace alo {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxLevels = 32;
constexpr int kMetaBytes = 640;  // H[32] | W[32] | start[32] as int32 | 1/W[32] 1/H[32] as float bits... see load_meta

template <typename CT>
struct alignas(16) FwdDesc {
    unsigned off[4];  // byte offset of each corner's (pixel, head 0, channel 0) inside the batch item's value slab
    CT w[4];          // bilinear corner weight * attention weight (0 for skipped samples)
};
template <typename CT>
struct alignas(16) BwdDesc {
    unsigned off[4];  // ELEMENT offset of each corner row, 0xFFFFFFFF = corner (or whole sample) not touched
    CT lh, lw, attn;
    int lvl;
};
constexpr unsigned kNoCorner = 0xFFFFFFFFu;

struct Dims {
    int S, M, D, L, P;
    int pairs_per_batch;   // Lq * M
    int blocks_per_batch;  // workgroups per batch item
    int iters_per_block;   // runs of (256 / G) pairs handled by one workgroup
    int runs_per_batch;    // wave kernel: runs of 16 pairs per batch item (head-major: ceil(Lq / 16) * M)
    unsigned nblocks;
    int ref_dim;           // fused prologue only: last dim of reference_points (2 or 4)
    int p_shift, m_shift;  // log2(P), log2(M) when they are powers of two, else -1 (integer division fallback)
    int loc_row_elems, attn_row_elems;  // fused wave kernels: elements between consecutive queries of the offsets / logits buffers
};

// Image-space position, validity and the four corners of one sampling point (cuh:285-291, :38-78).
template <typename CT>
struct Tap {
    bool valid, ok[4];
    int base;  // pixel index (within the batch item's S rows) of the (h_low, w_low) corner
    int W;
    CT lh, lw;
};
// MUL24: index arithmetic on the full-rate 24-bit integer multiplier (v_mul_lo_u32 is quarter rate); the caller
// guarantees S < 2^23 and row_bytes < 2^23.
// The float arithmetic of the sampling geometry is written with explicit fused multiply-adds and `fp contract(off)`: which
// products the compiler fuses otherwise depends on how the SLP vectoriser happened to pack the surrounding code, i.e. two
// kernels inlining the same source could disagree in the last bit of a sampling location — and kernels that promise
// bit-identical outputs (head-major, LDS-resident) would differ by a bf16 ulp on a few outputs per ten thousand.
__device__ __forceinline__ float fma_ct(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_ct(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename CT, bool MUL24 = false>
__device__ __forceinline__ Tap<CT> make_tap(CT loc_x, CT loc_y, int H, int W, int start) {
#pragma clang fp contract(off)
    Tap<CT> t;
    const CT h_im = fma_ct(loc_y, (CT)H, (CT)-0.5);
    const CT w_im = fma_ct(loc_x, (CT)W, (CT)-0.5);
    t.valid = (h_im > (CT)-1) && (w_im > (CT)-1) && (h_im < (CT)H) && (w_im < (CT)W);
    const CT hs = t.valid ? h_im : (CT)0, ws = t.valid ? w_im : (CT)0;  // keep the int conversion defined
    const CT hf = floor(hs), wf = floor(ws);
    const int h_low = (int)hf, w_low = (int)wf;
    t.lh = hs - hf;
    t.lw = ws - wf;
    const bool hl = h_low >= 0, hh = h_low + 1 <= H - 1, wl = w_low >= 0, wh = w_low + 1 <= W - 1;
    t.ok[0] = t.valid && hl && wl;
    t.ok[1] = t.valid && hl && wh;
    t.ok[2] = t.valid && hh && wl;
    t.ok[3] = t.valid && hh && wh;
    t.base = start + (MUL24 ? __mul24(h_low, W) : h_low * W) + w_low;
    t.W = W;
    return t;
}

__device__ __forceinline__ void load_meta(int* meta, const int32_t* shapes, const int32_t* lstart, int L) {
    const int t = threadIdx.x;
    if (t < L) {
        meta[t] = shapes[2 * t];
        meta[kMaxLevels + t] = shapes[2 * t + 1];
        meta[2 * kMaxLevels + t] = lstart[t];
        meta[3 * kMaxLevels + t] = (int)__float_as_uint(1.0f / (float)shapes[2 * t + 1]);  // 1 / W_l
        meta[4 * kMaxLevels + t] = (int)__float_as_uint(1.0f / (float)shapes[2 * t]);      // 1 / H_l
    }
}

// Sampling point -> gather descriptor: 4 corner byte offsets (out-of-range for corners that are not read) and 4 corner
// weights already multiplied by the attention weight.
template <typename CT, bool MUL24 = false>
__device__ __forceinline__ FwdDesc<CT> make_desc(CT x, CT y, CT a, int H, int W, int start, unsigned row_bytes) {
#pragma clang fp contract(off)
    FwdDesc<CT> d;
    const Tap<CT> t = make_tap<CT, MUL24>(x, y, H, W, start);
    const CT hh = (CT)1 - t.lh, hw = (CT)1 - t.lw;
    const CT w[4] = {hh * hw, hh * t.lw, t.lh * hw, t.lh * t.lw};
    // t.base may be negative (h_low = -1 with the lower corners still inside): signed multiply, wraps like the 32-bit one
    const unsigned o0 = MUL24 ? (unsigned)__mul24(t.base, (int)row_bytes) : (unsigned)t.base * row_bytes;
    const unsigned dy = MUL24 ? (unsigned)__mul24(t.W, (int)row_bytes) : (unsigned)t.W * row_bytes;
    const unsigned po[4] = {o0, o0 + row_bytes, o0 + dy, o0 + dy + row_bytes};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d.off[k] = t.ok[k] ? po[k] : kOutOfRange;
        d.w[k] = t.ok[k] ? w[k] * a : (CT)0;
    }
    return d;
}

// (x, y) of one sampling location / offset: one 4-byte load for bf16, one 8- / 16-byte load otherwise.
__device__ __forceinline__ void ld2(const float* p, float& x, float& y) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    x = v.x; y = v.y;
}
__device__ __forceinline__ void ld2(const double* p, double& x, double& y) {
    const double2 v = *reinterpret_cast<const double2*>(p);
    x = v.x; y = v.y;
}
__device__ __forceinline__ void ld2(const bf16_t* p, float& x, float& y) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    x = __uint_as_float(v << 16); y = __uint_as_float(v & 0xffff0000u);
}

// All-reduce over an aligned row of 16 lanes.  fp32: four DPP row rotations (pure VALU, no LDS round trips).
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ double row16_max(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double row16_sum(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// acc[0..VEC) += w * widen(row).  fp32 arithmetic is written on explicit 2-vectors so that every pair of channels is one
// v_pk_fma_f32 (left to itself the compiler packs three quarters of them and emits mul + mov + add for the rest).
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <typename Ld, typename CT, int VEC>
__device__ __forceinline__ void fma_row(const typename Ld::raw_t& raw, CT w, CT (&acc)[VEC]) {
    CT v[VEC];
    Ld::widen(raw, v);
    if constexpr (std::is_same<CT, float>::value && VEC % 2 == 0) {
        const f32x2 ww = {w, w};
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            const f32x2 r = __builtin_elementwise_fma(ww, f32x2{v[i], v[i + 1]}, f32x2{acc[i], acc[i + 1]});
            acc[i] = r.x;
            acc[i + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += w * v[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// FUSED = false: `loc` / `attn` are the reference op's inputs (sampling locations, softmax-ed attention weights), type LT.
// FUSED = true : `loc` holds the RAW sampling offsets and `attn` the RAW attention logits (both of value's type T) and
//                `ref` the reference points (N, Lq, L, ref_dim) of type CT; stage 1 evaluates what MSDeformAttn.forward
//                does between its linear layers and the op (ms_deform_attn.py:119-133): softmax over the L*P logits and
//                loc = ref + off / (W_l, H_l)   or   ref_xy + off / P * ref_wh * 0.5.
template <typename T, typename LT, typename CT, int VEC, int G, int LP_CT, int SB, bool FUSED>
__global__ void __launch_bounds__(kThreads)
msda_fwd_kernel(const T* __restrict__ value, const int32_t* __restrict__ shapes, const int32_t* __restrict__ lstart,
                const void* __restrict__ loc_, const void* __restrict__ attn_, const CT* __restrict__ ref,
                T* __restrict__ out, const Dims dm) {
    using InT = typename std::conditional<FUSED, T, LT>::type;
    const InT* loc = static_cast<const InT*>(loc_);
    const InT* attn = static_cast<const InT*>(attn_);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* meta = reinterpret_cast<int*>(smem);
    unsigned char* dbase = smem + kMetaBytes;
    using Desc = FwdDesc<CT>;
    using Ld = Loader<T, CT, VEC>;
    constexpr int PAIRS = kThreads / G;
    const int LP = LP_CT ? LP_CT : dm.L * dm.P;
    const int pair_stride = LP * (int)sizeof(Desc) + 16;  // +16 B: pairs of one wave land on distinct LDS slots

    const unsigned lb = xcd_contiguous_block(blockIdx.x, dm.nblocks);
    const int b = lb / dm.blocks_per_batch;
    const int chunk = lb % dm.blocks_per_batch;
    const int tid = threadIdx.x;

    load_meta(meta, shapes, lstart, dm.L);
    __syncthreads();

    const unsigned row_elems = (unsigned)dm.M * dm.D;
    const unsigned row_bytes = row_elems * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc =
        make_rsrc(value + (size_t)b * dm.S * row_elems, (unsigned)dm.S * row_bytes);
    const long batch_pair0 = (long)b * dm.pairs_per_batch;

    for (int it = 0; it < dm.iters_per_block; ++it) {
        const int pair0 = (chunk * dm.iters_per_block + it) * PAIRS;
        if (pair0 >= dm.pairs_per_batch) break;  // uniform

        // ---- stage 1: one descriptor per (pair, level, point) ------------------------------------------------------
        if constexpr (LP_CT == 16) {
            // L*P = 16: a thread keeps the same (level, point) slot s for every pair it serves, the 16 samples of a pair
            // sit in one aligned row of 16 lanes, and a thread serves NS pairs per run.  All global loads of the NS
            // samples are issued before the first is used (one memory latency per run, not 3*NS dependent ones);
            // indices of pairs past the end are clamped so no load is predicated, their descriptors are nulled below.
            constexpr int NS = PAIRS * 16 / kThreads;
            const int s = tid & 15;
            const int l = dm.p_shift >= 0 ? (s >> dm.p_shift) : s / dm.P;
            const int Hl = meta[l], Wl = meta[kMaxLevels + l], start = meta[2 * kMaxLevels + l];
            CT x[NS], y[NS], a[NS], r[NS][4];
            const int last_pair = dm.pairs_per_batch - 1;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int pair = min(pair0 + (tid >> 4) + j * (kThreads / 16), last_pair);
                const long g = (batch_pair0 + pair) * 16 + s;
                ld2(loc + 2 * g, x[j], y[j]);
                a[j] = (CT)ld(attn + g);
                if constexpr (FUSED) {
                    const int q = dm.m_shift >= 0 ? (pair >> dm.m_shift) : pair / dm.M;
                    const CT* rp = ref + (((long)b * (dm.pairs_per_batch / dm.M) + q) * dm.L + l) * dm.ref_dim;
                    r[j][0] = rp[0];
                    r[j][1] = rp[1];
                    if (dm.ref_dim == 4) { r[j][2] = rp[2]; r[j][3] = rp[3]; }  // uniform branch
                }
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int pl = (tid >> 4) + j * (kThreads / 16);
                CT xx = x[j], yy = y[j], aa = a[j];
                if constexpr (FUSED) {
                    // softmax over the pair's 16 logits, then loc = ref + off / (W_l, H_l)  |  ref_xy + off / P * ref_wh / 2.
                    // Storage narrower than fp32 (bf16) carries 2^-9 relative error in the logits and offsets themselves,
                    // so the hardware exp and reciprocal multiplies (<= 2 ulp) are used there; fp32 / fp64 keep the exact
                    // library exp and true divisions of the unfused path.
                    constexpr bool kFast = sizeof(T) < 4;
                    const CT mx = row16_max(aa);
                    if constexpr (kFast) aa = __expf(aa - mx); else aa = exp(aa - mx);
                    const CT sum = row16_sum(aa);
                    if constexpr (kFast) aa = aa * __builtin_amdgcn_rcpf(sum); else aa = aa / sum;
                    if (dm.ref_dim == 2) {
                        if constexpr (kFast) {
                            xx = r[j][0] + xx * __uint_as_float((unsigned)meta[3 * kMaxLevels + l]);
                            yy = r[j][1] + yy * __uint_as_float((unsigned)meta[4 * kMaxLevels + l]);
                        } else {
                            xx = r[j][0] + xx / (CT)Wl;
                            yy = r[j][1] + yy / (CT)Hl;
                        }
                    } else {
                        xx = r[j][0] + xx / (CT)dm.P * r[j][2] * (CT)0.5;
                        yy = r[j][1] + yy / (CT)dm.P * r[j][3] * (CT)0.5;
                    }
                }
                Desc d = make_desc<CT>(xx, yy, aa, Hl, Wl, start, row_bytes);
                if (pair0 + pl > last_pair) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { d.off[k] = kOutOfRange; d.w[k] = (CT)0; }
                }
                *reinterpret_cast<Desc*>(dbase + pl * pair_stride + s * (int)sizeof(Desc)) = d;
            }
        } else {
        const int nsamp = PAIRS * LP;
        for (int si = tid; si < nsamp; si += kThreads) {
            const int pl = si / LP, s = si - pl * LP;
            const int pair = pair0 + pl;
            Desc d;
#pragma unroll
            for (int k = 0; k < 4; ++k) { d.off[k] = kOutOfRange; d.w[k] = (CT)0; }
            const bool live = pair < dm.pairs_per_batch;
            const long g = (batch_pair0 + (live ? pair : 0)) * LP + s;
            CT x = (CT)0, y = (CT)0, a = (CT)0;
            const int l = dm.p_shift >= 0 ? (s >> dm.p_shift) : s / dm.P;  // wave-uniform choice, shift in the common case
            if (live) { x = (CT)ld(loc + 2 * g); y = (CT)ld(loc + 2 * g + 1); a = (CT)ld(attn + g); }
            if constexpr (FUSED) {
                const long g0 = g - s;
                CT mx = (CT)ld(attn + g0);
                for (int j = 1; j < LP; ++j) mx = fmax(mx, (CT)ld(attn + g0 + j));
                CT sum = (CT)0;
                for (int j = 0; j < LP; ++j) sum += exp((CT)ld(attn + g0 + j) - mx);
                a = exp(a - mx) / sum;
                if (live) {
                    const int q = dm.m_shift >= 0 ? (pair >> dm.m_shift) : pair / dm.M;
                    const CT* r = ref + (((long)b * (dm.pairs_per_batch / dm.M) + q) * dm.L + l) * dm.ref_dim;
                    if (dm.ref_dim == 2) {
                        x = r[0] + x / (CT)meta[kMaxLevels + l];
                        y = r[1] + y / (CT)meta[l];
                    } else {
                        x = r[0] + x / (CT)dm.P * r[2] * (CT)0.5;
                        y = r[1] + y / (CT)dm.P * r[3] * (CT)0.5;
                    }
                }
            }
            if (live) d = make_desc<CT>(x, y, a, meta[l], meta[kMaxLevels + l], meta[2 * kMaxLevels + l], row_bytes);
            *reinterpret_cast<Desc*>(dbase + pl * pair_stride + s * (int)sizeof(Desc)) = d;
        }
        }
        __syncthreads();

        // ---- stage 2: gather.  G lanes cover the channels of one pair ----------------------------------------------
        {
            const int pl = tid / G, lane = tid % G;
            const int pair = pair0 + pl;
            if (pair < dm.pairs_per_batch) {
                const int m = dm.m_shift >= 0 ? (pair & (dm.M - 1)) : pair % dm.M;
                const unsigned char* dp = dbase + pl * pair_stride;
                for (int c0 = lane * VEC; c0 < dm.D; c0 += G * VEC) {
                    const unsigned coff = (unsigned)(m * dm.D + c0) * (unsigned)sizeof(T);
                    CT acc[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = (CT)0;
                    // SB sampling points (4*SB corner rows) are requested before the first one is consumed; the
                    // outer loop stays rolled so the register allocator sees exactly that much in flight.
                    auto issue = [&](int s0, Desc (&d)[SB], typename Ld::raw_t (&raw)[SB][4]) {
#pragma unroll
                        for (int j = 0; j < SB; ++j) {
                            if (LP_CT || s0 + j < LP) {
                                d[j] = *reinterpret_cast<const Desc*>(dp + (s0 + j) * (int)sizeof(Desc));
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) { d[j].off[k] = kOutOfRange; d[j].w[k] = (CT)0; }
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) raw[j][k] = Ld::load(rsrc, d[j].off[k] + coff);
                        }
                    };
                    auto consume = [&](const Desc (&d)[SB], const typename Ld::raw_t (&raw)[SB][4]) {
#pragma unroll
                        for (int j = 0; j < SB; ++j) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) fma_row<Ld, CT, VEC>(raw[j][k], d[j].w[k], acc);
                        }
                    };
                    {
#pragma unroll 1
                        for (int s0 = 0; s0 < LP; s0 += SB) {
                            Desc d[SB];
                            typename Ld::raw_t raw[SB][4];
                            // requests at raised wave priority: they reach the memory system ahead of the other waves' FMA work
                            // (fp32 encoder call 0.451 -> 0.425 ms un-fused, 0.456 -> 0.445 fused; same box, alternating)
                            __builtin_amdgcn_s_setprio(3);
                            issue(s0, d, raw);
                            __builtin_amdgcn_s_setprio(0);
                            consume(d, raw);
                        }
                    }
                    store_vec<T, CT, VEC>(out + (batch_pair0 + pair) * dm.D + c0, acc);
                }
            }
        }
        __syncthreads();  // descriptors are rewritten by the next run
    }
}


// ------------------------------------------------------------------------------------------------------------------
// forward, bf16 values, L = P = 4, D <= 32 (the DETR-family configuration): wave-autonomous, weighted sum on the matrix pipe.
//
// What bounds the bf16 gather (measured, profiles/r01_pmc_counters.md): every (pair, corner) request is a 64-byte half
// of a 128-byte L1 line and the vector L1 serves one LINE per 2 clocks per CU — 91 M line reads per encoder launch =
// 0.29 ms at 2.4 GHz, whatever the kernel does around them.  The generic kernel sat at 0.31 ms with its VALU 78 % busy
// (1.5 instructions per gathered value: a shift/and to widen each bf16 + half a v_pk_fma_f32) and its two stages
// serialised by block barriers.  This kernel removes both side costs so that only the L1 line rate is left:
//   * v_mfma_f32_4x4x4_16B_bf16 multiplies 16 independent 4x4 blocks per wave — one block per (query, head) pair, i.e.
//     exactly the 4 lanes that serve a pair — consumes the bf16 rows as they come out of memory (no widening) and
//     accumulates in fp32:
//     D_pair[i][j] += sum_{k<4} A_pair[i][k] * B_pair[k][j]          k = the 4 corners of one sampling point
//     B[k][j]: lane j's channel of corner k (4 bf16 = one 64-bit operand; a 2-byte transpose of the loaded rows, v_perm)
//     A[i][k]: the corner weights.  They are fp32, the operand is bf16, so stage 1 splits every weight EXACTLY into three
//              bf16 terms w = hi + mid + lo (8 + 8 + 8 significant bits, truncation) and rows i = 0, 1, 2 carry one
//              term each (row 3 is zero).  Every product bf16 x bf16 is exact in fp32, so D[0] + D[1] + D[2] is the
//              fp32 weighted sum up to the order of the additions.
// Per sampling point a lane then issues 4 row loads, 16 v_perm and 8 MFMA instead of 32 widen + 16 v_pk_fma.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
struct alignas(16) MfmaDesc {
    unsigned off[4];      // corner byte offsets (kOutOfRange = not read)
    unsigned arow[4][2];  // A rows: {hi, mid, lo, 0} x 4 corners, bf16 pairs packed low-half-first
};
constexpr int kMfmaPairStride = 16 * (int)sizeof(MfmaDesc) + 16;

__device__ __forceinline__ s16x4 as_s16x4(unsigned lo, unsigned hi) {
    union { unsigned u[2]; s16x4 v; } x;
    x.u[0] = lo; x.u[1] = hi;
    return x.v;
}

__device__ __forceinline__ float quad_max(float v) {  // all-reduce over the 4 lanes of a quad, DPP quad_perm
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));  // [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));  // [2,3,0,1]
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
    return v;
}

// LDS operations of ONE wave execute in order, so data handed from lane to lane of a wave through LDS needs no wait at all — only
// the compiler must not move the reads above the writes.  The fence builtins at "wavefront" scope are NOT that: they lower to
// s_waitcnt vmcnt(0) lgkmcnt(0), i.e. every hand-over also waited for the wave's outstanding global loads, stores and atomics
// (the forward: its own output store and the next run's inputs; the tiled backward: all of a pass's row atomics).
#define ALO_WAVE_LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// One sampling point -> four corner byte offsets + four corner weights (already times the attention weight), the lean way the
// wave kernels use (fp32, `W * row_bytes` and `H * W` below 2^23).  Same semantics as make_desc (cuh:285-291, :38-78):
//   * the sample counts iff  h_im > -1 && w_im > -1 && h_im < H && w_im < W  (and the query exists: `alive`);
//   * a corner is read iff its row and its column are inside the map; every other corner gets the offset `oor`, which the
//     caller makes read as ZEROS without touching the map (past the buffer's range, or an all-zero LDS row).
// Unlike make_desc the weights of corners that are not read are NOT zeroed — they are finite and meet a zero row, which
// saves eight selects per sample (the descriptor stage is what bounds the LDS-resident kernel: VALU 88 % busy).  Weights are
// built separably: (1 - lh) * a and lh * a once per row, times (1 - lw) / lw per column.
// Addressing: pixel (h, w) of the level lives at base0 + (h * row_units + w * px_units) * unit_bytes — `row_units` / `px_units` are the
// row stride and the pixel pitch in units of `unit_bytes`, as floats.  The plain kernels use unit = one pixel (px_units = 1, folded
// away); the LDS-resident kernel pads every image row of a resident level by half a pixel (bank spreading), hence half-pixel units.
__device__ __forceinline__ void lean_sample(float x, float y, float a, bool alive, float Hf, float Wf, float row_units, float px_units,
                                            unsigned base0, unsigned unit_bytes, unsigned px_bytes, unsigned row_bytes_stride, unsigned oor,
                                            unsigned (&off)[4], float (&w)[4]) {
#pragma clang fp contract(off)
    const float h_im = __builtin_fmaf(y, Hf, -0.5f), w_im = __builtin_fmaf(x, Wf, -0.5f);
    const bool valid = alive && (h_im > -1.f) && (w_im > -1.f) && (h_im < Hf) && (w_im < Wf);
    const float hs = valid ? h_im : 0.f, ws = valid ? w_im : 0.f;   // keeps the arithmetic below finite and the conversions defined
    const float hf = floorf(hs), wf = floorf(ws);   // in [-1, H - 1] x [-1, W - 1] for a valid sample, 0 otherwise
    const float lh = hs - hf, lw = ws - wf;
    // row / column tests and the pixel index stay in fp32 (exact: |h * row_units + w * px_units| < 2^23): no conversions, no integer
    // multiply (v_mul_lo_u32 is quarter rate and the compiler reaches for it)
    const bool r0 = valid && hf >= 0.f, r1 = valid && hf < Hf - 1.f, c0 = wf >= 0.f, c1 = wf < Wf - 1.f;
    const float wy1 = lh * a, wy0 = (1.f - lh) * a, hw = 1.f - lw;
    w[0] = wy0 * hw;
    w[1] = wy0 * lw;
    w[2] = wy1 * hw;
    w[3] = wy1 * lw;
    // hf may be -1 with the lower corners still inside: the index is negative then and the byte offset wraps like the 32-bit one
    const int units = (int)__builtin_fmaf(hf, row_units, wf * px_units);
    const unsigned o0 = base0 + (unsigned)units * unit_bytes;
    const unsigned o2 = o0 + row_bytes_stride;
    off[0] = (r0 && c0) ? o0 : oor;
    off[1] = (r0 && c1) ? o0 + px_bytes : oor;
    off[2] = (r1 && c0) ? o2 : oor;
    off[3] = (r1 && c1) ? o2 + px_bytes : oor;
}

// MSDeformAttn's arithmetic between its linear layers and the op (ms_deform_attn.py:119-133) for the 4 points of ONE level of a
// (query, head) pair, bf16 storage: lane q of a quad holds level q.  `lr` = the 4 raw (x, y) offsets, `ar` = the 4 raw logits,
// r0..r3 = the reference point (r2, r3 used when ref_dim == 4).  Softmax over the pair's 16 logits (4 here, 12 in the other
// lanes of the quad); bf16 inputs carry 2^-9 relative error themselves, so the hardware exp / reciprocal (<= 2 ulp) are used.
// Shared by every wave kernel so that they agree bit for bit.
__device__ __forceinline__ void wave_prologue(const u32x4 lr, const u32x2 ar, float r0, float r1, float r2, float r3, int ref_dim,
                                              float inv_w, float inv_h, int P, float (&x)[4], float (&y)[4], float (&a)[4]) {
#pragma clang fp contract(off)
    const unsigned lw[4] = {lr.x, lr.y, lr.z, lr.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = __uint_as_float(lw[i] << 16);
        y[i] = __uint_as_float(lw[i] & 0xffff0000u);
    }
    a[0] = __uint_as_float(ar.x << 16); a[1] = __uint_as_float(ar.x & 0xffff0000u);
    a[2] = __uint_as_float(ar.y << 16); a[3] = __uint_as_float(ar.y & 0xffff0000u);
    const float mx = quad_max(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __expf(a[i] - mx);
    const float inv = __builtin_amdgcn_rcpf(quad_sum((a[0] + a[1]) + (a[2] + a[3])));
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] *= inv;
    // ONE wave-uniform branch around the four points (written inside the loop the compiler kept four diamonds, i.e. eight scalar
    // branches per run that also cut the descriptor stage into small scheduling regions)
    if (ref_dim == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = __builtin_fmaf(x[i], inv_w, r0);
            y[i] = __builtin_fmaf(y[i], inv_h, r1);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = r0 + x[i] / (float)P * r2 * 0.5f;
            y[i] = r1 + y[i] / (float)P * r3 * 0.5f;
        }
    }
}

// The four fp32 corner weights of a sample as the three A rows {hi, mid, lo} of the 4x4x4 MFMA: w = hi + mid + lo EXACTLY
// (8 + 8 + 8 significant bits by truncation; bf16(x) by truncation = the upper half of x, v_perm packs two upper halves).
__device__ __forceinline__ void split_weights(const float (&w)[4], u32x2 (&rows)[3]) {
#pragma clang fp contract(off)
    unsigned wb[4], r1b[4], r2b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wb[k] = __float_as_uint(w[k]);
        const float t1 = w[k] - __uint_as_float(wb[k] & 0xffff0000u);    // exact: the low 16 mantissa bits
        r1b[k] = __float_as_uint(t1);
        const float t2 = t1 - __uint_as_float(r1b[k] & 0xffff0000u);      // exact: <= 8 significant bits left
        r2b[k] = __float_as_uint(t2);
    }
    rows[0] = u32x2{__builtin_amdgcn_perm(wb[1], wb[0], 0x07060302u), __builtin_amdgcn_perm(wb[3], wb[2], 0x07060302u)};
    rows[1] = u32x2{__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
    rows[2] = u32x2{__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
}

// One WAVE is the unit of work (64-thread workgroups, no block barrier anywhere): its 16 quads serve 16 consecutive
// (query, head) pairs.  In stage 1 lane q of a quad turns the pair's 4 sampling points of level q into descriptors — its
// inputs are one 16-byte (offsets), one 8-byte (logits) and one 8/16-byte (reference point) load instead of 16 narrow
// ones, which matters because the texture path charges per instruction — and parks them in the wave's own LDS slice; in
// stage 2 the same quad gathers the pair's 64 corner rows.  Waves drift apart freely, so the VALU work of one wave's
// stage 1 overlaps the gathers of the others.  L = P = 4 and D <= 32 (the DETR-family configuration) only.
constexpr int kWaveLds = 16 * kMfmaPairStride;

// HM = true: `value` is HEAD-major, (N, M, S, D) — what alo_value_head_major writes — and a wave serves 16 CONSECUTIVE
// QUERIES of ONE head instead of 2 queries x 8 heads.  A head's row is D*2 = 64 bytes, half an L1 line: in the
// pixel-major layout every request of a wave instruction lands in its own line (16 line reads), in the head-major one
// neighbouring queries read neighbouring pixels of the same head, i.e. the same or the adjacent line.
template <int SB, bool FUSED, bool HM>
__global__ void __launch_bounds__(64)
msda_fwd_bf16_mfma_kernel(const bf16_t* __restrict__ value, const int32_t* __restrict__ shapes,
                          const int32_t* __restrict__ lstart, const void* __restrict__ loc_,
                          const void* __restrict__ attn_, const float* __restrict__ ref, bf16_t* __restrict__ out,
                          const Dims dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Ld = Loader<bf16_t, float, 8>;

    const unsigned lb = xcd_contiguous_block(blockIdx.x, dm.nblocks);
    const int b = lb / dm.blocks_per_batch;
    const int chunk = lb % dm.blocks_per_batch;
    const int pl = threadIdx.x >> 2, lane = threadIdx.x & 3;  // pair slot in the wave, lane in the quad (= level in stage 1)

    const unsigned row_elems = (unsigned)dm.M * dm.D;
    const unsigned row_bytes = HM ? (unsigned)dm.D * 2u : row_elems * 2u;  // bytes between neighbouring pixels of a head
    const long batch_pair0 = (long)b * dm.pairs_per_batch;
    const int last_pair = dm.pairs_per_batch - 1;
    const int Lq = dm.pairs_per_batch / dm.M;

    const int Hl = shapes[2 * lane], Wl = shapes[2 * lane + 1], start = lstart[lane];
    const float Hf = (float)Hl, Wf = (float)Wl;
    const float inv_w = 1.0f / Wf, inv_h = 1.0f / Hf;
    const unsigned base0 = (unsigned)start * row_bytes, w_bytes = (unsigned)Wl * row_bytes;
    unsigned char* dp = smem + pl * kMfmaPairStride;
    const int c0 = lane * 8;  // lanes with c0 >= D only feed the A rows

    for (int it = 0; it < dm.iters_per_block; ++it) {
        const int run = chunk * dm.iters_per_block + it;
        if (run >= dm.runs_per_batch) break;  // uniform
        int pair, m;
        bool dead;
        if constexpr (HM) {
            m = dm.m_shift >= 0 ? (run & (dm.M - 1)) : run % dm.M;
            const int q = (dm.m_shift >= 0 ? (run >> dm.m_shift) : run / dm.M) * 16 + pl;
            dead = q >= Lq;
            pair = min(q, Lq - 1) * dm.M + m;
        } else {
            pair = run * 16 + pl;
            dead = pair > last_pair;
            pair = min(pair, last_pair);
            m = dm.m_shift >= 0 ? (pair & (dm.M - 1)) : pair % dm.M;
        }
        const __amdgpu_buffer_rsrc_t rsrc =
            HM ? make_rsrc(value + ((size_t)b * dm.M + m) * dm.S * dm.D, (unsigned)dm.S * row_bytes)
               : make_rsrc(value + (size_t)b * dm.S * row_elems, (unsigned)dm.S * row_bytes);
        const long g0 = (batch_pair0 + pair) * 16 + 4 * lane;

        // ---- stage 1: the 4 points of level `lane` of this quad's pair ------------------------------------------------
        {
            float x[4], y[4], a[4];
            if constexpr (FUSED) {
                // offsets / logits rows may be slices of one wider buffer (a merged projection): row strides are arguments
                const int q = dm.m_shift >= 0 ? (pair >> dm.m_shift) : pair / dm.M;
                const long qrow = (long)b * Lq + q;
                const u32x4 lr = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(loc_) + qrow * dm.loc_row_elems + 32 * m + 8 * lane);
                const u32x2 ar = *reinterpret_cast<const u32x2*>(static_cast<const bf16_t*>(attn_) + qrow * dm.attn_row_elems + 16 * m + 4 * lane);
                const float* rp = ref + (((long)b * Lq + q) * dm.L + lane) * dm.ref_dim;
                float r0, r1, r2 = 0.f, r3 = 0.f;
                if (dm.ref_dim == 2) {
                    const float2 rv = *reinterpret_cast<const float2*>(rp);
                    r0 = rv.x; r1 = rv.y;
                } else {
                    const float4 rv = *reinterpret_cast<const float4*>(rp);
                    r0 = rv.x; r1 = rv.y; r2 = rv.z; r3 = rv.w;
                }
                wave_prologue(lr, ar, r0, r1, r2, r3, dm.ref_dim, inv_w, inv_h, dm.P, x, y, a);
            } else {
                const float* lp = static_cast<const float*>(loc_) + 2 * g0;
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(lp), l1 = *reinterpret_cast<const f32x4*>(lp + 4);
                const f32x4 av = *reinterpret_cast<const f32x4*>(static_cast<const float*>(attn_) + g0);
                x[0] = l0[0]; y[0] = l0[1]; x[1] = l0[2]; y[1] = l0[3];
                x[2] = l1[0]; y[2] = l1[1]; x[3] = l1[2]; y[3] = l1[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = av[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                MfmaDesc d;
                float w4[4];
                lean_sample(x[i], y[i], a[i], !dead, Hf, Wf, Wf, 1.0f, base0, row_bytes, row_bytes, w_bytes, kOutOfRange, d.off, w4);
                u32x2 rows[3];
                split_weights(w4, rows);
#pragma unroll
                for (int t = 0; t < 3; ++t) { d.arow[t][0] = rows[t].x; d.arow[t][1] = rows[t].y; }
                d.arow[3][0] = 0u;
                d.arow[3][1] = 0u;
                *reinterpret_cast<MfmaDesc*>(dp + (4 * lane + i) * (int)sizeof(MfmaDesc)) = d;
            }
        }
        // descriptors are exchanged inside the wave only: LDS operations of one wave execute in order, the fence keeps the
        // compiler from moving the reads below above the writes
        ALO_WAVE_LDS_ORDER();

        // ---- stage 2: gather + MFMA accumulate -----------------------------------------------------------------------
        {
            const unsigned coff = HM ? (unsigned)c0 * 2u : (unsigned)(m * dm.D + c0) * 2u;
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int s0 = 0; s0 < 16; s0 += SB) {
                u32x4 raw[SB][4];
                s16x4 arow[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const unsigned char* e = dp + (s0 + j) * (int)sizeof(MfmaDesc);
                    const u32x4 off = *reinterpret_cast<const u32x4*>(e);
                    const u32x2 ar = *reinterpret_cast<const u32x2*>(e + 16 + 8 * lane);
                    arow[j] = as_s16x4(ar.x, ar.y);
                    raw[j][0] = Ld::load(rsrc, off.x + coff);
                    raw[j][1] = Ld::load(rsrc, off.y + coff);
                    raw[j][2] = Ld::load(rsrc, off.z + coff);
                    raw[j][3] = Ld::load(rsrc, off.w + coff);
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned t0 = raw[j][0][q], t1 = raw[j][1][q], t2 = raw[j][2][q], t3 = raw[j][3][q];
                        const s16x4 b_even = as_s16x4(__builtin_amdgcn_perm(t1, t0, 0x05040100u),
                                                      __builtin_amdgcn_perm(t3, t2, 0x05040100u));
                        const s16x4 b_odd = as_s16x4(__builtin_amdgcn_perm(t1, t0, 0x07060302u),
                                                     __builtin_amdgcn_perm(t3, t2, 0x07060302u));
                        acc[2 * q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(arow[j], b_even, acc[2 * q], 0, 0, 0);
                        acc[2 * q + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(arow[j], b_odd, acc[2 * q + 1], 0, 0, 0);
                    }
                }
            }
            if (!dead && c0 < dm.D) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (acc[i][0] + acc[i][1]) + acc[i][2];
                store_vec<bf16_t, float, 8>(out + (batch_pair0 + pair) * dm.D + c0, o);
            }
        }
        ALO_WAVE_LDS_ORDER();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward, bf16, head-major, L = P = 4, D = 32: the COARSE pyramid levels of one (image, head) slab resident in LDS.
//
// What bounds msda_fwd_bf16_mfma_kernel: every corner row is a 64-byte request through the texture path, whose
// address / tag pipeline serves about one L1 line per two clocks per CU: stage 2 alone takes 0.22 ms for the 91 M rows of an
// encoder call and the descriptor arithmetic hides behind it.  Half of those rows belong to the two coarse levels, which are
// SMALL: levels 2 and 3 of the 1333 x 800 pyramid are 1050 + 273 pixels = 84.7 KB per (image, head) — they fit in a CU's 160 KB
// of LDS.
// So here one 12-wave workgroup per CU is pinned to a slab, copies the slab's coarse rows [res_row0, S) into LDS once (one
// coalesced pass, 22 MB over the whole launch), and its waves then serve every sample of a resident level from LDS and only the
// level-0 / level-1 samples through the buffer path.  The waves of a workgroup take runs of 16 consecutive queries round-robin,
// so at any time a CU works on ~190 consecutive queries of ONE head — a band of the image about one row wide whose level-0/1
// footprints overlap in L1.
//   * sample descriptors no longer travel through LDS whole: the lane that builds a sample keeps its four corner addresses in
//     registers and the other lanes of the quad read them with DPP quad broadcasts folded into the address add (v_add_u32_dpp);
//     only the A rows of the MFMA (the three bf16 terms of the corner weights, 24 bytes per sample) go through the wave's LDS
//     slice, which is the transposition "row i of every sample to lane i" — 6 KB per wave instead of 12.5.
//   * resident rows are consumed with ds_read_b64_tr_b16, the LDS transpose read: in every 16-lane group lane 4 r + p fetches
//     eight bytes of corner r of pair p, and lane 4 p + j receives element j of the four corners — which IS the K-vector of the
//     4x4x4 MFMA for one channel (the buffer path needs 16 v_perm per sample for that 2-byte transposition: the kernel is VALU
//     bound, so they are what the resident half no longer pays).  For lane j to receive "its" channel 8 j + n from the n-th read,
//     a resident row is stored channel-permuted: position 4 n + j holds channel 8 j + n.  The fetching lane learns the corner's
//     address from a 2 KB table the sample's owner writes over the (by then consumed) A rows of the buffer-path samples.
//   * corners outside the map aim at an all-zero LDS row (resident levels) or past the slab (buffer bounds check): no branches,
//     and no byte outside the sampled footprint is ever read (NaN-safe like the other kernels).
//   * RL = 2 is the first resident level.  `res_row0` comes from the HOST's copy of the level starts; the kernel compares it
//     with the device copy and, should they disagree, serves every level through the buffer path: the hint steers speed, never
//     results.  (A level-3-only variant for pyramids whose level 2 does not fit was built and measured 10-20 % SLOWER than the
//     plain head-major kernel at 1600 x 1200 — a quarter of the samples does not pay for the 12-wave / 168-register shape — and
//     was removed; such launches take the plain kernel.)
// ------------------------------------------------------------------------------------------------------------------
constexpr int kResWaves = 12;                            // 3 waves per SIMD: <= 168 registers
constexpr int kResThreads = 64 * kResWaves;
constexpr int kResSampleStride = 16 * 24;                // bytes between the A rows of consecutive samples (16 pairs x 24 B)
constexpr int kResTable = 8 * 256;                       // corner addresses of up to 8 resident samples x 16 pairs x 4 corners
constexpr int kResWaveLds = 16 * kResSampleStride;       // 6144 B per wave (the table takes over consumed A rows)
static_assert(kResTable <= 8 * kResSampleStride, "the table must fit in the A rows of the buffer-path samples");
constexpr int kResLdsTotal = 160 * 1024;
constexpr int kResRowPad = 8;                            // bytes appended to every image row of a resident level (see below)
constexpr int kResFixed = 64 /* zero row */ + 16;
constexpr int kResMaxImage = kResLdsTotal - kResWaves * kResWaveLds - kResFixed;
static_assert(kResFixed % 16 == 0 && kResWaveLds % 16 == 0, "work areas behind the (16-byte rounded) image stay 16-byte aligned");   // 90 032 bytes for the resident image
typedef short v4i16_t __attribute__((ext_vector_type(4)));

struct ResDims {
    int res_row0;       // first resident row of a slab (host copy of level_start_index[2])
    int res_rows;       // S - res_row0
    int h[2], w[2];     // host copy of the shapes of the resident levels 2 and 3
    int image_bytes;    // LDS bytes of the resident image = sum over resident levels of H * (W * 64 + kResRowPad)
    int wps;            // workgroups per (image, head) slab
    int runs_per_slab;  // ceil(Lq / 16)
    int runs_per_wg;    // ceil(runs_per_slab / wps)
};

template <int J>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {   // value of lane J of every quad (v_mov_dpp quad_perm:[J,J,J,J])
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xf, 0xf, false);
}

__global__ void __launch_bounds__(kResThreads)
msda_fwd_bf16_resident_kernel(const bf16_t* __restrict__ value, const int32_t* __restrict__ shapes,
                              const int32_t* __restrict__ lstart, const void* __restrict__ loc_,
                              const void* __restrict__ attn_, const float* __restrict__ ref, bf16_t* __restrict__ out,
                              const Dims dm, const ResDims rd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Ld = Loader<bf16_t, float, 8>;
    constexpr int RL = 2;   // first resident level

    const unsigned lb = xcd_contiguous_block(blockIdx.x, dm.nblocks);
    const int slab = (int)(lb / (unsigned)rd.wps), part = (int)(lb % (unsigned)rd.wps);
    const int b = slab / dm.M, m = slab - b * dm.M;
    const int lid = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lid >> 2, lane = lid & 3;   // pair slot in the wave, lane in the quad (= level in stage 1)

    const bf16_t* slab_base = value + ((size_t)b * dm.M + m) * dm.S * 32;
    const unsigned zero_row = (unsigned)rd.image_bytes;                           // LDS byte address of the all-zero row
    unsigned char* aw = smem + zero_row + kResFixed + wave * kResWaveLds;         // this wave's A rows ...
    unsigned char* tab = aw;   // ... whose first rows, once consumed, take the table of the resident samples' corner addresses

    // Resident image: level l (H x W) is stored row by row at a 64-byte pixel pitch, every image row followed by kResRowPad = 8
    // bytes.  With nothing between the rows the four corners of a sample and the samples of neighbouring queries would share four
    // bank groups (8-way conflicts on the transpose reads: 0.285 ms); the pad moves the rows y and y + 1 two banks apart.  A 72-byte
    // PIXEL pitch spreads a little better (16 M instead of 27 M conflict cycles per launch) but costs 10 KB more LDS, i.e. the twelfth
    // wave; row pads of 8 / 16 / 32 bytes and the 72-byte pitch all measure 0.177-0.180 ms.
    const unsigned rs0 = (unsigned)rd.w[0] * 64u + kResRowPad, rs1 = (unsigned)rd.w[1] * 64u + kResRowPad;
    const unsigned lds_lvl1 = (unsigned)rd.h[0] * rs0;   // where the second resident level starts

    // the host's view of the resident levels must be the device's; otherwise nothing is treated as resident (wave-uniform)
    bool res_ok = rd.res_row0 == lstart[RL] && shapes[2 * RL] == rd.h[0] && shapes[2 * RL + 1] == rd.w[0];
    res_ok = res_ok && lstart[3] == rd.res_row0 + rd.h[0] * rd.w[0] && shapes[6] == rd.h[1] && shapes[7] == rd.w[1];
    res_ok = res_ok && rd.res_row0 + rd.h[0] * rd.w[0] + rd.h[1] * rd.w[1] == dm.S;

    // ---- the slab's coarse rows -> LDS (one coalesced pass), zero row ---------------------------------------------------------
    {
        // channel-permuted image: a 16-byte granule holds channels 8 j .. 8 j + 7 of its pixel; channel 8 j + n goes to position 4 n + j
        const u32x4* src = reinterpret_cast<const u32x4*>(slab_base + (size_t)rd.res_row0 * 32);
        const int ngran = rd.res_rows * 4, n0 = rd.h[0] * rd.w[0];
        for (int g = threadIdx.x; g < ngran; g += kResThreads) {
            const u32x4 v = src[g];
            int r = g >> 2;
            const bool second = r >= n0;
            r -= second ? n0 : 0;
            const int wl = second ? rd.w[1] : rd.w[0];
            const int y = r / wl, x = r - y * wl;
            const unsigned at = (second ? lds_lvl1 : 0u) + (unsigned)y * (second ? rs1 : rs0) + (unsigned)x * 64u;
            unsigned short* px = reinterpret_cast<unsigned short*>(smem + at) + (g & 3);
#pragma unroll
            for (int n = 0; n < 8; ++n) px[4 * n] = (unsigned short)(v[n >> 1] >> (16 * (n & 1)));
        }
        if (threadIdx.x < 16) reinterpret_cast<unsigned*>(smem + zero_row)[threadIdx.x] = 0u;
    }
    __syncthreads();

    const unsigned row_bytes = 64u;
    const int Lq = dm.pairs_per_batch / dm.M;
    // wave-uniform bases of this image's rows + 32-bit per-lane element offsets: the loads take an SGPR base and one VGPR offset
    const bf16_t* loc_b = static_cast<const bf16_t*>(loc_) + (size_t)b * Lq * dm.loc_row_elems;
    const bf16_t* attn_b = static_cast<const bf16_t*>(attn_) + (size_t)b * Lq * dm.attn_row_elems;
    const float* ref_b = ref + (size_t)b * Lq * 4 * dm.ref_dim;
    bf16_t* out_b = out + (size_t)b * dm.pairs_per_batch * 32;
    const unsigned lane_loc = 32u * m + 8u * lane, lane_attn = 16u * m + 4u * lane, lane_out = 32u * m + 8u * lane;
    const int Hl = shapes[2 * lane], Wl = shapes[2 * lane + 1];
    const bool lane_res = res_ok && lane >= RL;
    const unsigned oor = lane_res ? zero_row : kOutOfRange;
    const float Hf = (float)Hl, Wf = (float)Wl;
    const float inv_w = 1.0f / Wf, inv_h = 1.0f / Hf;
    // addressing in units of kResRowPad bytes: a pixel is 64 / pad units, an image row W * 64 / pad units in the slab and one more in LDS
    constexpr float kPxUnits = 64.0f / (float)kResRowPad;
    const float row_units = lane_res ? kPxUnits * Wf + 1.0f : kPxUnits * Wf;
    const unsigned w_bytes = (unsigned)Wl * 64u + (lane_res ? (unsigned)kResRowPad : 0u);
    const unsigned base0 = lane_res ? (lane == RL ? 0u : lds_lvl1) : (unsigned)lstart[lane] * row_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(slab_base, (unsigned)dm.S * row_bytes);
    const unsigned coff = (unsigned)lane * 16u;   // this lane's 8 channels of a 64-byte row
    // transpose-read roles: in its 16-lane group this lane FETCHES corner (lid >> 2) & 3 of the group's pair lid & 3
    const unsigned fetch_off = (unsigned)((((lid >> 4) * 4 + (lid & 3)) * 16) + ((lid >> 2) & 3) * 4);
    const int run_hi = rd.runs_per_slab;
    unsigned char* aw_wr = aw + (4 * lane) * kResSampleStride + pl * 24;
    const unsigned char* aw_rd = aw + pl * 24 + 8 * min(lane, 2);   // lane 3's row of A is ignored (D[3] is never summed)

    // The raw offsets / logits / reference point of a run are requested ONE RUN AHEAD (under the previous run's gathers): with two
    // waves per SIMD a run that starts by waiting for its own inputs leaves the SIMD idle for a memory round trip.
    struct RunIn {
        int run;
        bool dead;
        unsigned qc;
        u32x4 lr;
        u32x2 ar;
        f32x4 rv;
    };
    int next_index = wave;   // runs are dealt out round-robin over the workgroup's waves (every run costs the same instructions)
    auto next_run = [&]() -> RunIn {
        RunIn in;
        // The workgroups of a slab take every wps-th run and their waves every twelfth of those: the slab's 12 * wps waves sweep
        // the image TOGETHER, so the level-0 / 1 rows they gather at any moment come from a band of a few image rows (and the eight
        // heads of an image, which share an XCD, read the same offset / logit lines at the same time) instead of from wps
        // separate quarters of the pyramid: survey 0.196 -> 0.184 ms, ring 0.1695 -> 0.166, uniform unchanged.
        in.run = part + rd.wps * next_index;
        next_index += kResWaves;
        const int q = in.run * 16 + pl;
        in.dead = q >= Lq || in.run >= run_hi;
        in.qc = (unsigned)min(q, Lq - 1);   // row of this query inside the image: 32-bit element offsets (host-checked)
        in.lr = *reinterpret_cast<const u32x4*>(loc_b + (in.qc * (unsigned)dm.loc_row_elems + lane_loc));
        in.ar = *reinterpret_cast<const u32x2*>(attn_b + (in.qc * (unsigned)dm.attn_row_elems + lane_attn));
        const float* rp = ref_b + (in.qc * 4u + (unsigned)lane) * (unsigned)dm.ref_dim;
        if (dm.ref_dim == 2) {
            const float2 v = *reinterpret_cast<const float2*>(rp);
            in.rv = f32x4{v.x, v.y, 0.f, 0.f};
        } else {
            in.rv = *reinterpret_cast<const f32x4*>(rp);
        }
        return in;
    };
    RunIn cur = next_run();
    while (cur.run < run_hi) {   // wave-uniform
        const bool dead = cur.dead;
        const unsigned qc = cur.qc;

        // ---- stage 1: the 4 points of level `lane` of this quad's (query, head) pair --------------------------------------------
        unsigned off[4][4];
        {
            float x[4], y[4], a[4];
            const u32x4 lr = cur.lr;
            const u32x2 ar = cur.ar;
            const float r0 = cur.rv[0], r1 = cur.rv[1], r2 = cur.rv[2], r3 = cur.rv[3];
            wave_prologue(lr, ar, r0, r1, r2, r3, dm.ref_dim, inv_w, inv_h, dm.P, x, y, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float w4[4];
                lean_sample(x[i], y[i], a[i], !dead, Hf, Wf, row_units, kPxUnits, base0, (unsigned)kResRowPad, row_bytes, w_bytes, oor, off[i], w4);
                u32x2 rows[3];
                split_weights(w4, rows);
                u32x2* dst = reinterpret_cast<u32x2*>(aw_wr + i * kResSampleStride);
                dst[0] = rows[0];
                dst[1] = rows[1];
                dst[2] = rows[2];
            }
        }
        // the next run's inputs are requested NOW, into the registers stage 1 has just finished with: they travel under this run's
        // gathers (a run index past the end clamps to the image's last query: the loads stay in bounds)
        const RunIn nxt = next_run();
        // A rows are exchanged inside the wave only: LDS operations of one wave execute in order, the fence keeps the compiler
        // from moving the reads below above the writes
        ALO_WAVE_LDS_ORDER();

        // ---- stage 2: gather + MFMA accumulate; group J = the 4 samples of level J, built by lane J of the quad -----------------
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto consume = [&](const u32x4 (&raw)[4][4], const s16x4 (&arow)[4]) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned t0 = raw[p][0][c], t1 = raw[p][1][c], t2 = raw[p][2][c], t3 = raw[p][3][c];
                    const s16x4 b_even = as_s16x4(__builtin_amdgcn_perm(t1, t0, 0x05040100u), __builtin_amdgcn_perm(t3, t2, 0x05040100u));
                    const s16x4 b_odd = as_s16x4(__builtin_amdgcn_perm(t1, t0, 0x07060302u), __builtin_amdgcn_perm(t3, t2, 0x07060302u));
                    acc[2 * c] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(arow[p], b_even, acc[2 * c], 0, 0, 0);
                    acc[2 * c + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(arow[p], b_odd, acc[2 * c + 1], 0, 0, 0);
                }
            }
        };
        // buffer path: group J's corner addresses come from lane J of the quad (DPP broadcast folded into the add)
#define ALO_RES_ISSUE(J, RAW)                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                         \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                         \
            RAW[p][k] = Ld::load(rsrc, quad_bcast<J>(off[p][k]) + coff);
#define ALO_RES_CONSUME(J, RAW)                                                                                                \
        {                                                                                                                      \
            s16x4 arow[4];                                                                                                     \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                                   \
                const u32x2 ar = *reinterpret_cast<const u32x2*>(aw_rd + (4 * J + p) * kResSampleStride);                     \
                arow[p] = as_s16x4(ar.x, ar.y);                                                                                \
            }                                                                                                                  \
            consume(RAW, arow);                                                                                                \
        }
        // LDS path: the fetching lane reads its corner's address from the table, eight transpose reads deliver the eight B operands
        // of a sample.  The reads of sample t + 1 are issued BEFORE the MFMAs of sample t (left to itself the compiler keeps three
        // reads in flight and every MFMA waits most of an LDS round trip: 2.5 k cycles per run).
        auto resident_samples = [&]() {
            constexpr int NS = 4 * (4 - RL);
            unsigned ca[NS];
            s16x4 arow[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                ca[t] = *reinterpret_cast<const unsigned*>(tab + t * 256 + fetch_off);
                const u32x2 ar = *reinterpret_cast<const u32x2*>(aw_rd + (4 * RL + t) * kResSampleStride);
                arow[t] = as_s16x4(ar.x, ar.y);
            }
            v4i16_t bt[2][8];
#pragma unroll
            for (int n = 0; n < 8; ++n)
                bt[0][n] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)(smem + ca[0] + 8 * n));
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (t + 1 < NS) {
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        bt[(t + 1) & 1][n] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4i16_t*)(smem + ca[t + 1] + 8 * n));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(arow[t], bt[t & 1][n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Order of the sum: levels 0, 1, 2, 3, as in the other wave kernels (same products, same order: bit-identical outputs).
        if (res_ok) {   // wave-uniform
            // the sixteen row requests of a group go out at raised wave priority (s_setprio): a wave that has just finished its
            // descriptors gets its loads into the memory system ahead of its two neighbours' permutes and MFMAs instead of
            // interleaving with them — 0.1757-0.1773 -> 0.1705-0.1712 ms, same box, alternating (raising it for the whole buffer
            // part, or from stage 1 on, measures the same; for everything but the LDS part less)
            {
                u32x4 raw[4][4];
                __builtin_amdgcn_s_setprio(3);
                ALO_RES_ISSUE(0, raw)
                __builtin_amdgcn_s_setprio(0);
                ALO_RES_CONSUME(0, raw)
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                u32x4 raw[4][4];
                __builtin_amdgcn_s_setprio(3);
                ALO_RES_ISSUE(1, raw)
                __builtin_amdgcn_s_setprio(0);
                ALO_RES_CONSUME(1, raw)
            }
            __builtin_amdgcn_sched_barrier(0);
            ALO_WAVE_LDS_ORDER();
            if (lane >= RL) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    *reinterpret_cast<u32x4*>(tab + ((lane - RL) * 4 + p) * 256 + pl * 16) = u32x4{off[p][0], off[p][1], off[p][2], off[p][3]};
            }
            ALO_WAVE_LDS_ORDER();
            resident_samples();
        } else {   // the host's view of the pyramid is not the device's: every level through the buffer path
            u32x4 raw[4][4];
            ALO_RES_ISSUE(0, raw)
            ALO_RES_CONSUME(0, raw)
            __builtin_amdgcn_sched_barrier(0);
            ALO_RES_ISSUE(1, raw)
            ALO_RES_CONSUME(1, raw)
            __builtin_amdgcn_sched_barrier(0);
            ALO_RES_ISSUE(2, raw)
            ALO_RES_CONSUME(2, raw)
            __builtin_amdgcn_sched_barrier(0);
            ALO_RES_ISSUE(3, raw)
            ALO_RES_CONSUME(3, raw)
        }
#undef ALO_RES_ISSUE
#undef ALO_RES_CONSUME
        if (!dead) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (acc[i][0] + acc[i][1]) + acc[i][2];
            store_vec<bf16_t, float, 8>(out_b + (qc * (unsigned)dm.M * 32u + lane_out), o);
        }
        cur = nxt;
        ALO_WAVE_LDS_ORDER();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
template <typename T, typename LT, typename CT, int VEC, int G, int LP_CT>
__global__ void __launch_bounds__(kThreads)
msda_bwd_kernel(const T* __restrict__ value, const int32_t* __restrict__ shapes, const int32_t* __restrict__ lstart,
                const LT* __restrict__ loc, const LT* __restrict__ attn, const T* __restrict__ grad_out,
                CT* __restrict__ grad_value, CT* __restrict__ grad_loc, CT* __restrict__ grad_attn, const Dims dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* meta = reinterpret_cast<int*>(smem);
    unsigned char* dbase = smem + kMetaBytes;
    using Desc = BwdDesc<CT>;
    using Ld = Loader<T, CT, VEC>;
    constexpr int PAIRS = kThreads / G;
    const int LP = LP_CT ? LP_CT : dm.L * dm.P;
    const int pair_stride = LP * (int)sizeof(Desc) + 16;

    const unsigned lb = xcd_contiguous_block(blockIdx.x, dm.nblocks);
    const int b = lb / dm.blocks_per_batch;
    const int chunk = lb % dm.blocks_per_batch;
    const int tid = threadIdx.x;

    load_meta(meta, shapes, lstart, dm.L);
    __syncthreads();

    const unsigned row_elems = (unsigned)dm.M * dm.D;
    const size_t slab = (size_t)b * dm.S * row_elems;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(value + slab, (unsigned)dm.S * row_elems * (unsigned)sizeof(T));
    CT* gv = grad_value + slab;
    const long batch_pair0 = (long)b * dm.pairs_per_batch;

    for (int it = 0; it < dm.iters_per_block; ++it) {
        const int pair0 = (chunk * dm.iters_per_block + it) * PAIRS;
        if (pair0 >= dm.pairs_per_batch) break;

        const int nsamp = PAIRS * LP;
        for (int si = tid; si < nsamp; si += kThreads) {
            const int pl = si / LP, s = si - pl * LP;
            const int pair = pair0 + pl;
            Desc d;
#pragma unroll
            for (int k = 0; k < 4; ++k) d.off[k] = kNoCorner;
            d.lh = d.lw = d.attn = (CT)0;
            d.lvl = 0;
            if (pair < dm.pairs_per_batch) {
                const int l = s / dm.P;
                const long g = (batch_pair0 + pair) * LP + s;
                const CT x = (CT)ld(loc + 2 * g), y = (CT)ld(loc + 2 * g + 1);
                const Tap<CT> t = make_tap<CT>(x, y, meta[l], meta[kMaxLevels + l], meta[2 * kMaxLevels + l]);
                const int px[4] = {t.base, t.base + 1, t.base + t.W, t.base + t.W + 1};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t.ok[k]) d.off[k] = (unsigned)px[k] * row_elems;
                d.lh = t.lh;
                d.lw = t.lw;
                d.attn = (CT)ld(attn + g);
                d.lvl = t.valid ? l : -1;
            }
            *reinterpret_cast<Desc*>(dbase + pl * pair_stride + s * (int)sizeof(Desc)) = d;
        }
        __syncthreads();

        {
            const int pl = tid / G, lane = tid % G;
            const int pair = pair0 + pl;
            const bool live = pair < dm.pairs_per_batch;
            const int m = live ? pair % dm.M : 0;
            const unsigned char* dp = dbase + pl * pair_stride;
            const T* go = grad_out + (batch_pair0 + (live ? pair : 0)) * dm.D;
            // Whole waves walk the samples together (the shuffles below need every lane of a group present).
            for (int s = 0; s < LP; ++s) {
                const Desc d = *reinterpret_cast<const Desc*>(dp + s * (int)sizeof(Desc));
                CT s_attn = (CT)0, s_w = (CT)0, s_h = (CT)0;
                if (live && d.lvl >= 0) {
                    const CT hh = (CT)1 - d.lh, hw = (CT)1 - d.lw;
                    const CT w[4] = {hh * hw, hh * d.lw, d.lh * hw, d.lh * d.lw};
                    for (int c0 = lane * VEC; c0 < dm.D; c0 += G * VEC) {
                        const unsigned ch = (unsigned)(m * dm.D + c0);
                        typename Ld::raw_t raw[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned o = d.off[k] == kNoCorner ? kOutOfRange : (d.off[k] + ch) * (unsigned)sizeof(T);
                            raw[k] = Ld::load(rsrc, o);
                        }
                        CT v[4][VEC];
#pragma unroll
                        for (int k = 0; k < 4; ++k) Ld::widen(raw[k], v[k]);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const CT top = (CT)ld(go + c0 + i);
                            const CT tgv = top * d.attn;
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (d.off[k] != kNoCorner) unsafeAtomicAdd(gv + d.off[k] + ch + i, w[k] * tgv);
                            const CT val = w[0] * v[0][i] + w[1] * v[1][i] + w[2] * v[2][i] + w[3] * v[3][i];
                            const CT ghw = -hw * v[0][i] - d.lw * v[1][i] + hw * v[2][i] + d.lw * v[3][i];
                            const CT gww = -hh * v[0][i] + hh * v[1][i] - d.lh * v[2][i] + d.lh * v[3][i];
                            s_attn += top * val;
                            s_w += gww * tgv;
                            s_h += ghw * tgv;
                        }
                    }
                }
                // channel sums of the group: fp32 groups of 16 / 32 lanes reduce on DPP (row rotations + one row broadcast,
                // pure VALU), everything else through wave shuffles; `writer` is a lane that ends up with the total
                int writer = 0;
                if constexpr (std::is_same<CT, float>::value && (G == 32 || G == 16)) {
                    s_attn = row16_sum(s_attn);
                    s_w = row16_sum(s_w);
                    s_h = row16_sum(s_h);
                    if constexpr (G == 32) {  // second row of the group += lane 15 of its first row
                        s_attn += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s_attn), 0x142, 0xa, 0xf, false));
                        s_w += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s_w), 0x142, 0xa, 0xf, false));
                        s_h += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s_h), 0x142, 0xa, 0xf, false));
                        writer = 16;
                    }
                } else {
#pragma unroll
                    for (int off = G / 2; off > 0; off >>= 1) {
                        s_attn += __shfl_xor(s_attn, off, 64);
                        s_w += __shfl_xor(s_w, off, 64);
                        s_h += __shfl_xor(s_h, off, 64);
                    }
                }
                if (live && lane == writer) {
                    const long g = (batch_pair0 + pair) * LP + s;
                    CT W = (CT)0, H = (CT)0;
                    if (d.lvl >= 0) { H = (CT)meta[d.lvl]; W = (CT)meta[kMaxLevels + d.lvl]; }
                    grad_attn[g] = s_attn;
                    grad_loc[2 * g] = W * s_w;
                    grad_loc[2 * g + 1] = H * s_h;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, fp32, D = 32, L = P = 4 (the DETR-family shape): tiled, window-dense, on the matrix cores.
//
// msda_bwd_kernel above scatters one 128-byte atomic row per (query, head, point, corner): 45.5 M rows per encoder-size
// call, and the memory-side atomic units retire ~16 G rows/s — 2.9 of its 4.1 ms.  Queries that are neighbours in the image
// sample overlapping pixels, so the sums can be formed on chip first.  Here ONE WAVE (a 64-thread workgroup, no block barrier
// anywhere) owns a TILE of 16 queries of one head — a 4x4 block of one pyramid level when the queries are the pyramid's own
// pixels (Lq == S: the encoder's self-attention), 16 consecutive queries otherwise:
//   1. lane (query i, level l) turns its 4 sampling points into taps; the wave reduces the per-level bounding box of all its
//      taps (the WINDOW), stacks the windows of the levels that fit together into one range of rows of its LDS region (with a
//      table of where every row lives in the frame) and writes the scalar weights w_corner * attn into a dense matrix
//      A[row][16 queries]: plain read-add-write, no LDS atomics — lane (i, l) is the only writer of column i of level l's rows.
//   2. grad_value of the stacked rows is the dense product  dV[row][ch] = sum_q A[row][q] * grad_out[q][ch]  (32 rows x 32
//      channels per MFMA tile, K = the 16 queries): ONE atomic row per touched window pixel per tile instead of one per corner —
//      about 1/7 of the per-corner count; buffer atomics whose lanes aim past the slab where there is nothing to add.
//   3. the channel dot products every sample needs, d[pixel][q] = <value[pixel], grad_out[q]>, are the transposed dense
//      product over the same window, the value rows going from memory straight into the matrix operand; D lands in the wave's
//      LDS region in place of A and lane (i, l) picks its 16 corner values from there:
//      grad_attn = sum_k w_k d_k,  grad_loc = attn * (W, H) * (...).
// Both products run on the fp32 matrix instructions (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32 FMA chains).
// Tried and dropped (DESIGN.md 4.2 has the table): 4-wave workgroups that share a walk over the common bounding box of four tiles
// (half the atomic rows, but two block barriers per pass); three-term bf16 and two-term fp16 splits of the operands on the 16-bit
// MFMAs (the matrix chain is not on the critical path); smaller LDS regions for 3-4 waves per SIMD (the register file spills
// first); persistent waves, statically strided or fed from a queue (slower than one tile per hardware-dispatched workgroup).
// The wave's LDS region holds 300 window pixels; the levels are made resident in as many passes as it takes.  A level whose
// window does not fit (queries of a coarse level looking at a fine one, decoder queries, arbitrary locations) takes the
// per-corner route of msda_bwd_kernel for that level only: same results, old cost.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kPxBudget = 300;                      // window pixels resident in LDS at a time (64 B each): 8 waves per CU
constexpr int kWaveRegion = kPxBudget * 64;         // bytes
constexpr int kBatch = 6;                           // 16-row blocks of value rows in flight in stage 3 (4: 3 % slower)
constexpr unsigned kDropped = 0xffffff00u;             // a byte offset past any frame slab: the buffer range check drops the lane
constexpr int kRowTable = (kPxBudget + 31) / 32 * 32;   // byte offset (inside the frame's slab) of every resident window row
constexpr int kTileLds = kWaveRegion + kRowTable * 4;   // 20480 bytes: 8 waves per CU

struct TileDims {
    int S, M, Lq;
    int pyramid;        // 1: Lq == S and the queries are tiled as 4x4 blocks of their level; 0: 16 consecutive queries
    int n_super;        // tiles per batch item (pyramid: sum over levels of ceil(H/4) * ceil(W/4), from the host's copy of the shapes)
    unsigned nblocks;
};

typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// All-reduce of a packed (x, y) pair of int16 over the 16 lanes that share (lane & 3): two DPP row rotations, two shuffles.
template <bool MIN>
__device__ __forceinline__ int same_level_pk(int v) {
    auto op = [](int a, int b) {
        const s16x2 x = __builtin_bit_cast(s16x2, a), y = __builtin_bit_cast(s16x2, b);
        return __builtin_bit_cast(int, MIN ? __builtin_elementwise_min(x, y) : __builtin_elementwise_max(x, y));
    };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = op(v, __shfl_xor(v, 16, 64));
    v = op(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {   // sum over each 32-lane half; valid in lanes 16-31 / 48-63
    v = row16_sum(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    return v;
}

__global__ void __launch_bounds__(64, 2)
msda_bwd_tiled_kernel(const float* __restrict__ value, const int32_t* __restrict__ shapes, const int32_t* __restrict__ lstart,
                      const float* __restrict__ loc, const float* __restrict__ attn, const float* __restrict__ grad_out,
                      float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
                      const TileDims td) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    float* region = reinterpret_cast<float*>(smem);
    unsigned* rowoff = reinterpret_cast<unsigned*>(smem + kWaveRegion);               // [kRowTable]

    const unsigned lb = xcd_contiguous_block(blockIdx.x, td.nblocks);
    const int m = lb % td.M;
    // tiles in reverse order: those of the coarse levels (whose queries look at wide windows of the fine levels and take
    // the per-corner route there) start first and overlap the many light ones instead of forming the tail
    const int st = td.n_super - 1 - (int)((lb / td.M) % td.n_super);
    const int b = lb / (td.M * td.n_super);
    const int M = td.M, S = td.S, Lq = td.Lq;

    // ---- which queries --------------------------------------------------------------------------------------------------
    // The 4x4 tiling of every level is derived HERE from the device copy of the shapes (the one the arithmetic below uses); the
    // host only sized the grid.  A tile index past the device-side tile count has no queries (query_of() returns -1).
    int lq = 0, t0q = 0, twq = 1;
    bool no_tile = false;
    bool pyramid = td.pyramid != 0;
    if (pyramid) {
        int t0 = 0;
        lq = -1;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int th = (shapes[2 * l] + 3) >> 2, tw = (shapes[2 * l + 1] + 3) >> 2;
            if (lq < 0 && st < t0 + th * tw) { lq = l; t0q = t0; twq = tw; }
            t0 += th * tw;
        }
        // A host copy that UNDER-counts the device's tiles (same S, other shapes: (5, 8) against (2, 20)) would leave the last
        // device tiles without a workgroup, i.e. queries whose grad_loc / grad_attn rows are never written.  Every workgroup sees
        // the same two numbers, so the whole launch then groups the queries 16 in a row instead (the host's grid always holds
        // ceil(Lq / 16) super-tiles: sum ceil(h/4) ceil(w/4) >= S / 16): slower windows, same results.
        if (t0 > td.n_super) { pyramid = false; lq = 0; t0q = 0; twq = 1; }
        no_tile = lq < 0;
        if (no_tile) lq = 0;
    }
    const int trow = pyramid ? (st - t0q) / twq : 0;
    const int tcol = pyramid ? (st - t0q) - trow * twq : 0;
    const int Hq = no_tile ? 0 : shapes[2 * lq], Wq = shapes[2 * lq + 1], Sq = lstart[lq];
    auto query_of = [&](int i) -> int {   // slot i of the tile -> query index, -1 past the edge
        if (pyramid) {
            const int qy = trow * 4 + (i >> 2), qx = tcol * 4 + (i & 3);
            return (qy < Hq && qx < Wq) ? Sq + qy * Wq + qx : -1;
        }
        const int q = st * 16 + i;
        return q < Lq ? q : -1;
    };
    const long bq0 = (long)b * Lq;

    // ---- stage 1: taps and windows ------------------------------------------------------------------------------------------
    const int qi = lane >> 2, lev = lane & 3;
    const int q_own = query_of(qi);
    const bool live = q_own >= 0;
    const int Hl = shapes[2 * lev], Wl = shapes[2 * lev + 1];
    const long qm = (bq0 + (live ? q_own : 0)) * M + m;
    float lh[4], lw[4], at[4];
    int h_low[4], w_low[4];
    unsigned flags[4];   // bit k: corner k is inside the map; bit 4: the sample is valid (cuh:285-291, :38-78)
    const float* lp = loc + (qm * 4 + lev) * 8;
    const f32x4 l0 = *reinterpret_cast<const f32x4*>(lp), l1 = *reinterpret_cast<const f32x4*>(lp + 4);
    const f32x4 av = *reinterpret_cast<const f32x4*>(attn + (qm * 4 + lev) * 4);
    // grad_out rows of all four sub-tiles as B operands of the 32x32x2 product: lane (kg = lane >> 5, ch = lane & 31) holds
    // G_j[8 kg + s][ch], s = 0..7.  Issued behind the (smaller) location loads: their latency hides behind the tap arithmetic.
    float G[8];
    {
        const float* gb = grad_out + (bq0 * M + m) * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int q = query_of(8 * (lane >> 5) + s);
            G[s] = q >= 0 ? gb[(unsigned)(q * M) * 32u] : 0.f;
        }
    }
    {
        const float xs[4] = {l0[0], l0[2], l1[0], l1[2]}, ys[4] = {l0[1], l0[3], l1[1], l1[3]};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float h_im = ys[p] * (float)Hl - 0.5f, w_im = xs[p] * (float)Wl - 0.5f;
            const bool valid = live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
            const float hs = valid ? h_im : 0.f, ws = valid ? w_im : 0.f;
            const float hf = floorf(hs), wf = floorf(ws);
            h_low[p] = (int)hf;
            w_low[p] = (int)wf;
            lh[p] = hs - hf;
            lw[p] = ws - wf;
            at[p] = av[p];
            const bool hl = h_low[p] >= 0, hh = h_low[p] + 1 <= Hl - 1, wl = w_low[p] >= 0, wh = w_low[p] + 1 <= Wl - 1;
            flags[p] = valid ? ((hl && wl ? 1u : 0u) | (hl && wh ? 2u : 0u) | (hh && wl ? 4u : 0u) | (hh && wh ? 8u : 0u) | 16u) : 0u;
        }
    }
    const bool small_map = Hl < 32768 && Wl < 32768;   // window coordinates travel as packed int16
    int x0 = 32767, x1 = -32768, y0 = 32767, y1 = -32768;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if ((flags[p] & 16u) && small_map) {   // a valid sample has w_low in [-1, W-1]: a column (row) of its footprint is inside
            x0 = min(x0, max(w_low[p], 0));
            x1 = max(x1, min(w_low[p] + 1, Wl - 1));
            y0 = min(y0, max(h_low[p], 0));
            y1 = max(y1, min(h_low[p] + 1, Hl - 1));
        }
    {
        const int lo = same_level_pk<true>((x0 & 0xffff) | (y0 << 16));
        const int hi = same_level_pk<false>((x1 & 0xffff) | (y1 << 16));
        x0 = (int)(short)(lo & 0xffff); y0 = lo >> 16;
        x1 = (int)(short)(hi & 0xffff); y1 = hi >> 16;
    }
    const int ww = x1 >= x0 ? x1 - x0 + 1 : 0, wh = x1 >= x0 ? y1 - y0 + 1 : 0;
    // a level without a window: no valid sample (nothing to do) or a map too large for the packed coordinates (per-corner route)
    const long np_l = (long)ww * wh;
    int vmask = (int)((flags[0] | flags[1] | flags[2] | flags[3]) >> 4);   // any valid sample on this level in the sub-tile?
    vmask |= __builtin_amdgcn_update_dpp(0, vmask, 0x124, 0xf, 0xf, false);
    vmask |= __builtin_amdgcn_update_dpp(0, vmask, 0x128, 0xf, 0xf, false);
    vmask |= __shfl_xor(vmask, 16, 64);
    vmask |= __shfl_xor(vmask, 32, 64);
    const int np_own = (np_l > kPxBudget || (vmask && !small_map)) ? kPxBudget + 1 : (int)np_l;
    // residency plan: levels in order, as many passes over the wave's region as it takes (wave-uniform)
    int np4[4], off4[4], pass4[4];   // pass4: 0..3 = resident in that pass, 8 = per-corner route, 9 = nothing to do
    int my_npass = 1;
    {
        int used = 0, pass = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            np4[l] = __builtin_amdgcn_readlane(np_own, l);
            off4[l] = 0;
            if (np4[l] == 0) { pass4[l] = 9; continue; }
            if (np4[l] > kPxBudget) { pass4[l] = 8; continue; }
            if (used + np4[l] > kPxBudget) { ++pass; used = 0; }
            pass4[l] = pass; off4[l] = used; used += np4[l];
            my_npass = pass + 1;
        }
    }
    const int my_pass = lev == 0 ? pass4[0] : (lev == 1 ? pass4[1] : (lev == 2 ? pass4[2] : pass4[3]));
    const int off = lev == 0 ? off4[0] : (lev == 1 ? off4[1] : (lev == 2 ? off4[2] : off4[3]));
    const bool any_corner_route = pass4[0] == 8 || pass4[1] == 8 || pass4[2] == 8 || pass4[3] == 8;
    int base[4];   // window row of the (h_low, w_low) corner; the others are +1, +ww, +ww+1
#pragma unroll
    for (int p = 0; p < 4; ++p) base[p] = (h_low[p] - y0) * ww + (w_low[p] - x0);

    // own grad_out rows as the B operand of stage 3 (16x16x4): lane (kg = lane >> 4, q = lane & 15) holds g[q][16 c + 4 kg + s];
    // the A operand uses the same (c, kg, s) -> channel map, so every channel meets itself
    f32x4 g4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    {
        const int qb = query_of(lane & 15);
        if (qb >= 0) {
            const float* gp = grad_out + ((bq0 + qb) * M + m) * 32 + 4 * (lane >> 4);
            g4[0] = *reinterpret_cast<const f32x4*>(gp);
            g4[1] = *reinterpret_cast<const f32x4*>(gp + 16);
        }
    }

    // wave-uniform slab bases + 32-bit per-lane byte offsets: the address arithmetic of every gather / atomic is one v_mad
    char* gv_b = reinterpret_cast<char*>(grad_value + (size_t)b * S * M * 32 + m * 32);
    const char* vb3 = reinterpret_cast<const char*>(value + (size_t)b * S * M * 32 + m * 32);
    const __amdgpu_buffer_rsrc_t gv_rsrc = __builtin_amdgcn_make_buffer_rsrc(gv_b, 0, (int)((unsigned)(S * M - m) * 128u), 0x00020000);
    const unsigned gv_lane = (unsigned)(lane & 31) * 4u, v_lane = (unsigned)(lane >> 4) * 16u, row_b = (unsigned)M * 128u;
    // window geometry of the four levels as wave-uniform scalars
    int sx0[4], sy0[4], sww[4], sW[4], sS[4];
    float sinv[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        sx0[l] = __builtin_amdgcn_readlane(x0, l);
        sy0[l] = __builtin_amdgcn_readlane(y0, l);
        sww[l] = __builtin_amdgcn_readlane(ww, l);
        sW[l] = shapes[2 * l + 1];
        sS[l] = lstart[l];
        // (r + 0.5) * sinv is at least 0.5 / ww away from an integer (ww <= kPxBudget): the rounding of rcp cannot move its floor
        sinv[l] = __builtin_amdgcn_rcpf((float)max(sww[l], 1));
    }
    for (int pass = 0; pass < my_npass; ++pass) {
        // ---- stage 1b: A of this pass's levels; the levels' windows are stacked in the region, rows [off4[l], off4[l] + np4[l]) -----
        const bool mine = my_pass == pass;
        int used = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (pass4[l] == pass) used = off4[l] + np4[l];
        const int used32 = (used + 31) & ~31;
        for (int o = lane * 4; o < used * 16; o += 256) *reinterpret_cast<f32x4*>(region + o) = f32x4{0.f, 0.f, 0.f, 0.f};
        // where every stacked row lives in the frame (a byte offset inside the (b, head) slab); the tail of the last 32-row block
        // gets an offset the buffer range check drops
        for (int r = lane; r < used32; r += 64) {
            int lo = 0, wx = 0, wy = 0, wwl = 1, Wv = 0, Sv = 0;
            float inv = 1.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in_k = pass4[k] == pass && r >= off4[k];   // levels are stacked in order: the last match owns the row
                lo = in_k ? off4[k] : lo;
                wx = in_k ? sx0[k] : wx;
                wy = in_k ? sy0[k] : wy;
                wwl = in_k ? sww[k] : wwl;
                Wv = in_k ? sW[k] : Wv;
                Sv = in_k ? sS[k] : Sv;
                inv = in_k ? sinv[k] : inv;
            }
            const int rl = r - lo;
            const int ry = (int)(((float)rl + 0.5f) * inv), rx = rl - ry * wwl;
            rowoff[r] = r < used ? (unsigned)(Sv + (wy + ry) * Wv + wx + rx) * row_b : kDropped;
        }
        ALO_WAVE_LDS_ORDER();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (mine && (flags[p] & 16u)) {
                const float hh = 1.f - lh[p], hw = 1.f - lw[p];
                const float w4[4] = {hh * hw * at[p], hh * lw[p] * at[p], lh[p] * hw * at[p], lh[p] * lw[p] * at[p]};
                const int idx[4] = {base[p], base[p] + 1, base[p] + ww, base[p] + ww + 1};
                float cur[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[k] = (flags[p] >> k) & 1u ? region[(off + idx[k]) * 16 + qi] : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((flags[p] >> k) & 1u) region[(off + idx[k]) * 16 + qi] = cur[k] + w4[k];
            }
        }
        ALO_WAVE_LDS_ORDER();

        // ---- stage 2: grad_value of the stacked rows, 32 at a time -----------------------------------------------------------------
        // (rows of the last block past `used` multiply whatever the region holds there: each output row depends on its own A row
        // only, and theirs are dropped)
        for (int blk = 0; blk < (used32 >> 5); ++blk) {
            const float* ap = region + (blk * 32 + (lane & 31)) * 16 + 8 * (lane >> 5);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
            u32x4 po[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) po[g] = *reinterpret_cast<const u32x4*>(rowoff + blk * 32 + 8 * g + 4 * (lane >> 5));
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], G[e], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], G[4 + e], acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // no exec juggling: lanes with nothing to add (untouched rows hold exact zeros) aim past the slab
                    const float a = acc[4 * g + e];
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(a, gv_rsrc, a != 0.f ? (po[g][e] | gv_lane) : kDropped, 0, 0);
                }
        }

        // ---- stage 3: d[row][q] = <value[row], grad_out[q]> over the stacked rows, then the per-sample gradients -------------------
        for (int blk0 = 0; blk0 < ((used + 15) >> 4); blk0 += kBatch) {
            f32x4 v[kBatch][2];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int r = (blk0 + u) * 16 + (lane & 15);
                v[u][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                v[u][1] = v[u][0];
                if (r < used) {
                    const float* vp = reinterpret_cast<const float*>(vb3 + (rowoff[r] + v_lane));
                    v[u][0] = *reinterpret_cast<const f32x4*>(vp);
                    v[u][1] = *reinterpret_cast<const f32x4*>(vp + 16);
                }
            }
            // (stage 2 has read every A row before the first D row lands: one wave, LDS in order)
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if ((blk0 + u) * 16 >= used) break;   // wave-uniform
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) d = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][c][e], g4[c][e], d, 0, 0, 0);
                const int row0 = (blk0 + u) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (row0 + reg < used) region[(row0 + reg) * 16 + (lane & 15)] = d[reg];
            }
        }
        ALO_WAVE_LDS_ORDER();
        if (mine && live) {
            f32x4 ga, gl0, gl1;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float gx = 0.f, gy = 0.f, gat = 0.f;
                if (flags[p] & 16u) {
                    const int idx[4] = {base[p], base[p] + 1, base[p] + ww, base[p] + ww + 1};
                    float dk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) dk[k] = (flags[p] >> k) & 1u ? region[(off + idx[k]) * 16 + qi] : 0.f;
                    const float hh = 1.f - lh[p], hw = 1.f - lw[p];
                    gat = hh * hw * dk[0] + hh * lw[p] * dk[1] + lh[p] * hw * dk[2] + lh[p] * lw[p] * dk[3];
                    const float gww = -hh * dk[0] + hh * dk[1] - lh[p] * dk[2] + lh[p] * dk[3];
                    const float ghw = -hw * dk[0] - lw[p] * dk[1] + hw * dk[2] + lw[p] * dk[3];
                    gx = (float)Wl * gww * at[p];
                    gy = (float)Hl * ghw * at[p];
                }
                ga[p] = gat;
                if (p < 2) { gl0[2 * p] = gx; gl0[2 * p + 1] = gy; } else { gl1[2 * p - 4] = gx; gl1[2 * p - 3] = gy; }
            }
            float* glp = grad_loc + (qm * 4 + lev) * 8;
            *reinterpret_cast<f32x4*>(glp) = gl0;
            *reinterpret_cast<f32x4*>(glp + 4) = gl1;
            *reinterpret_cast<f32x4*>(grad_attn + (qm * 4 + lev) * 4) = ga;
        }
        ALO_WAVE_LDS_ORDER();
    }
    // levels with nothing to do (no valid sample of the sub-tile): their gradients are zero
    if (my_pass == 9 && live) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        float* glp = grad_loc + (qm * 4 + lev) * 8;
        *reinterpret_cast<f32x4*>(glp) = z;
        *reinterpret_cast<f32x4*>(glp + 4) = z;
        *reinterpret_cast<f32x4*>(grad_attn + (qm * 4 + lev) * 4) = z;
    }

    // ---- stage 4: levels that are not resident: one 128-byte row per corner, as msda_bwd_kernel does ---------------------------------
    if (!any_corner_route) return;   // wave-uniform
    // sample descriptors through the wave's region (free now): 8 words per (level, point, query), read back as broadcasts
    {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4* dst = reinterpret_cast<u32x4*>(region) + ((lev * 4 + p) * 16 + qi) * 2;
            dst[0] = u32x4{flags[p], (unsigned)q_own, (unsigned)h_low[p], (unsigned)w_low[p]};
            dst[1] = u32x4{__float_as_uint(lh[p]), __float_as_uint(lw[p]), __float_as_uint(at[p]), 0u};
        }
        ALO_WAVE_LDS_ORDER();
    }
#pragma unroll
    for (int lv = 0; lv < 4; ++lv) {
        if (pass4[lv] != 8) continue;   // wave-uniform
        const int Hlv = shapes[2 * lv], Wlv = shapes[2 * lv + 1], Slv = lstart[lv];
        const int ch = lane & 31, half = lane >> 5;
        const float* vb = value + (size_t)b * S * M * 32 + m * 32 + ch;
        float* gvb = grad_value + (size_t)b * S * M * 32 + m * 32 + ch;
#pragma unroll 2
        for (int it = 0; it < 32; ++it) {   // 4 points x 16 queries, one sample per half-wave per step
            const int p = it >> 3, i = 2 * (it & 7) + half;
            const u32x4* src = reinterpret_cast<const u32x4*>(region) + ((lv * 4 + p) * 16 + i) * 2;
            const u32x4 d0 = src[0], d1 = src[1];
            const unsigned f = d0.x;
            const int q = (int)d0.y, hl_ = (int)d0.z, wl_ = (int)d0.w;
            const float lh_ = __uint_as_float(d1.x), lw_ = __uint_as_float(d1.y), at_ = __uint_as_float(d1.z);
            float s_attn = 0.f, s_w = 0.f, s_h = 0.f;
            if (f & 16u) {
                const float top = grad_out[((bq0 + q) * M + m) * 32 + ch];
                const float tgv = top * at_;
                const float hh = 1.f - lh_, hw = 1.f - lw_;
                const float w4[4] = {hh * hw, hh * lw_, lh_ * hw, lh_ * lw_};
                const long pix0 = (long)Slv + (long)hl_ * Wlv + wl_;
                const long px[4] = {pix0, pix0 + 1, pix0 + Wlv, pix0 + Wlv + 1};
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (f >> k) & 1u ? vb[(size_t)px[k] * M * 32] : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((f >> k) & 1u) unsafeAtomicAdd(gvb + (size_t)px[k] * M * 32, w4[k] * tgv);
                s_attn = top * (w4[0] * v[0] + w4[1] * v[1] + w4[2] * v[2] + w4[3] * v[3]);
                s_w = tgv * (-hh * v[0] + hh * v[1] - lh_ * v[2] + lh_ * v[3]);
                s_h = tgv * (-hw * v[0] - lw_ * v[1] + hw * v[2] + lw_ * v[3]);
            }
            s_attn = half_wave_sum(s_attn);
            s_w = half_wave_sum(s_w);
            s_h = half_wave_sum(s_h);
            if (ch == 16 && q >= 0) {
                const long g = (((bq0 + q) * M + m) * 4 + lv) * 4 + p;
                grad_attn[g] = s_attn;
                grad_loc[2 * g] = (float)Wlv * s_w;
                grad_loc[2 * g + 1] = (float)Hlv * s_h;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------------------------
struct Plan {
    int vec, g;
    bool lp16;
};

inline int pick_group(int lanes_needed) {
    static const int kGroups[] = {4, 8, 16, 64};
    for (int g : kGroups)
        if (lanes_needed <= g) return g;
    return 64;
}

Plan make_plan(int D, int L, int P, size_t elem, bool aligned16) {
    Plan p;
    const int vec = (int)(16 / elem);
    if (aligned16 && D % vec == 0) {
        p.vec = vec;
        p.g = pick_group(D / vec);
    } else {
        p.vec = 1;
        p.g = D <= 8 ? 8 : 64;
    }
    p.lp16 = (L * P == 16) && p.vec != 1 && p.g != 64;
    return p;
}

Dims make_dims(int N, int S, int M, int D, int L, int Lq, int P, int G, long target_blocks = 4096) {
    Dims d;
    d.S = S; d.M = M; d.D = D; d.L = L; d.P = P;
    d.pairs_per_batch = Lq * M;
    d.ref_dim = 0;
    d.loc_row_elems = M * L * P * 2;
    d.attn_row_elems = M * L * P;
    d.p_shift = (P & (P - 1)) == 0 ? __builtin_ctz((unsigned)P) : -1;
    d.m_shift = (M & (M - 1)) == 0 ? __builtin_ctz((unsigned)M) : -1;
    const int pairs = kThreads / G;
    const long iters_total = ((long)d.pairs_per_batch + pairs - 1) / pairs;
    long ipb = iters_total * N / target_blocks;  // keep >= ~4096 workgroups of 4 waves in flight when the problem allows it
    if (ipb < 1) ipb = 1;
    if (ipb > 4) ipb = 4;   // workgroups in flight sweep the image in query order: the shorter their shares, the narrower the band of rows
                            // the L2 has to hold (8 -> 4: fp32 survey 0.497 -> 0.473 ms, plain head-major 0.224 -> 0.216; 2 and 1 lose to
                            // the per-workgroup overhead)
    d.iters_per_block = (int)ipb;
    d.runs_per_batch = (int)iters_total;
    d.blocks_per_batch = (int)((iters_total + ipb - 1) / ipb);
    d.nblocks = (unsigned)(d.blocks_per_batch * N);
    return d;
}

template <typename K>
int launch(K kernel, const Dims& dm, size_t lds, hipStream_t stream, const char* what, void** args, int threads = kThreads) {
    if (lds > 160 * 1024) return fail(ALO_ERR_UNSUPPORTED, "%s: L*P too large for LDS (%zu bytes)", what, lds);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    }
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), dim3(dm.nblocks), dim3(threads), args, lds,
                                   stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return check_launch(what);
}

#define ALO_FWD_CASE(T, LT, CT, VEC, G, LPCT)                                                                     \
    if (plan.vec == VEC && plan.g == G && plan.lp16 == (LPCT == 16)) {                                             \
        const size_t lds = kMetaBytes + (size_t)(kThreads / G) * ((size_t)L * P * sizeof(FwdDesc<CT>) + 16);       \
        if (fused)                                                                                                 \
            return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, (LPCT ? 4 : 2), true>, dm, lds, stream, "alo_msda_forward_fused", args); \
        return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, (LPCT ? 4 : 2), false>, dm, lds, stream, "alo_msda_forward", args);     \
    }
#define ALO_BWD_CASE(T, LT, CT, VEC, G, LPCT)                                                                     \
    if (plan.vec == VEC && plan.g == G && plan.lp16 == (LPCT == 16)) {                                             \
        const size_t lds = kMetaBytes + (size_t)(kThreads / G) * ((size_t)L * P * sizeof(BwdDesc<CT>) + 16);       \
        return launch(msda_bwd_kernel<T, LT, CT, VEC, G, LPCT>, dm, lds, stream, "alo_msda_backward", args);       \
    }
// every (vector width, group) pair a plan can produce for one dtype
#define ALO_ALL_CASES(CASE, T, LT, CT, VECW)                                                       \
    CASE(T, LT, CT, VECW, 4, 16) CASE(T, LT, CT, VECW, 8, 16) CASE(T, LT, CT, VECW, 16, 16)         \
    CASE(T, LT, CT, VECW, 4, 0) CASE(T, LT, CT, VECW, 8, 0) CASE(T, LT, CT, VECW, 16, 0)            \
    CASE(T, LT, CT, VECW, 64, 0) CASE(T, LT, CT, 1, 8, 0) CASE(T, LT, CT, 1, 32, 0) CASE(T, LT, CT, 1, 64, 0)

int validate(const void* value, const int32_t* shapes, const int32_t* lstart, const void* loc, const void* attn,
             int N, int S, int M, int D, int L, int Lq, int P, int vdt, int ldt, size_t* elem_out) {
    ALO_REQUIRE(value && shapes && lstart && loc && attn, ALO_ERR_INVALID_ARGUMENT, "msda: null pointer argument");
    ALO_REQUIRE(N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0, ALO_ERR_INVALID_ARGUMENT,
                "msda: dimensions must be positive (N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)", N, S, M, D, L, Lq, P);
    ALO_REQUIRE(L <= kMaxLevels, ALO_ERR_UNSUPPORTED, "msda: at most %d levels are supported, got %d", kMaxLevels, L);
    const bool ok = (vdt == ALO_F32 && ldt == ALO_F32) || (vdt == ALO_F64 && ldt == ALO_F64) ||
                    (vdt == ALO_BF16 && ldt == ALO_F32);
    ALO_REQUIRE(ok, ALO_ERR_UNSUPPORTED, "msda: unsupported dtype pair (value=%d, loc=%d)", vdt, ldt);
    const size_t elem = vdt == ALO_F64 ? 8 : (vdt == ALO_F32 ? 4 : 2);
    ALO_REQUIRE((double)S * M * D * elem < 3.0 * 1024 * 1024 * 1024, ALO_ERR_UNSUPPORTED,
                "msda: one batch item of value must stay below 3 GiB");
    ALO_REQUIRE((double)Lq * M < 2.0e9 && (double)Lq * M * L * P < 9.0e18, ALO_ERR_UNSUPPORTED, "msda: Lq*M too large");
    *elem_out = elem;
    return ALO_OK;
}

}  // namespace
}  // namespace alo

using namespace alo;

namespace {
// Whether msda_fwd_bf16_resident_kernel serves a launch (levels 2 and 3 of every slab resident in LDS), from the HOST's copy of the
// shapes: returns 2 (the first resident level) or 0 (the plain head-major kernel serves the launch).  Fills `rd` when non-zero.
// policy ALO_RESIDENT_AUTO: only where the resident kernel is the faster one; ALO_RESIDENT_ALWAYS: wherever it can run.
int resident_plan(const int32_t* host_shapes, int N, int S, int M, int L, int Lq, int policy, ResDims* rd) {
    if (L != 4) return 0;
    long start[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 4; ++l) {
        const long h = host_shapes[2 * l], w = host_shapes[2 * l + 1];
        if (h <= 0 || w <= 0 || h * (w * 64 + kResRowPad) / kResRowPad >= (1L << 23)) return 0;   // offsets in pad units must stay exact in fp32
        start[l + 1] = start[l] + h * w;
    }
    if (start[4] != S) return 0;
    long bytes = 0;
    for (int l = 2; l < 4; ++l) bytes += (long)host_shapes[2 * l] * (host_shapes[2 * l + 1] * 64L + kResRowPad);
    // the zero row and the waves' work areas follow the image: keep them 16-byte aligned (the image itself is only 8-byte
    // granular when H2 + H3 is odd, and the kernel writes its tables with 16-byte stores)
    bytes = (bytes + 15) & ~15L;
    if (bytes > kResMaxImage) return 0;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const long slabs = (long)N * M;
    rd->runs_per_slab = (Lq + 15) / 16;
    // Measured over frame sizes 256 x 320 ... 1333 x 800 and N = 1 ... 8 (tools/exp/res_sweep.py, HIP-graph replay): the resident kernel
    // beats the plain one as soon as every wave of the chip gets a run (>= CUs x 12 runs in the launch; 0.77-0.86 of the plain kernel's
    // time from there on, 1.2-1.6 x below), and splitting a slab over more workgroups — down to ONE run per wave — is never slower than
    // fewer, longer workgroups: the copy of the coarse rows is cheap next to an idle CU.
    if (policy != ALO_RESIDENT_ALWAYS && slabs * rd->runs_per_slab < (long)cus * kResWaves) return 0;
    long wps = slabs >= cus ? 1 : (cus + slabs - 1) / slabs;
    const long wps_cap = rd->runs_per_slab / kResWaves;   // at least one run per wave
    if (wps > wps_cap) wps = wps_cap;
    if (wps < 1 || slabs * wps >= 0x7fffffffL) return 0;
    rd->res_row0 = (int)start[2];
    rd->res_rows = (int)(S - start[2]);
    rd->h[0] = host_shapes[4]; rd->w[0] = host_shapes[5];
    rd->h[1] = host_shapes[6]; rd->w[1] = host_shapes[7];
    rd->image_bytes = (int)bytes;
    rd->wps = (int)wps;
    rd->runs_per_wg = (int)((rd->runs_per_slab + wps - 1) / wps);
    return 2;
}

int forward_impl(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index, const void* loc,
                 const void* attn, const void* ref, int ref_dim, void* out, int N, int S, int M, int D, int L, int Lq,
                 int P, int value_dtype, int loc_dtype, void* stream_, bool head_major = false, long loc_row_elems = 0,
                 long attn_row_elems = 0, const int32_t* host_shapes = nullptr, int resident_policy = ALO_RESIDENT_AUTO) {
    size_t elem = 0;
    if (int rc = validate(value, spatial_shapes, level_start_index, loc, attn, N, S, M, D, L, Lq, P, value_dtype,
                          loc_dtype, &elem))
        return rc;
    ALO_REQUIRE(out, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward: out is null");
    const bool fused = ref != nullptr;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool aligned = (((uintptr_t)value | (uintptr_t)out) & 15) == 0;
    const Plan plan = make_plan(D, L, P, elem, aligned);
    Dims dm = make_dims(N, S, M, D, L, Lq, P, plan.g);
    dm.ref_dim = ref_dim;
    void* args[] = {&value, &spatial_shapes, &level_start_index, &loc, &attn, &ref, &out, &dm};
    const bool in_aligned = (((uintptr_t)loc | (uintptr_t)attn | (uintptr_t)(fused ? ref : nullptr)) & 15) == 0;
    const bool wave_kernel = value_dtype == ALO_BF16 && aligned && in_aligned && L == 4 && P == 4 && D % 8 == 0 &&
                             D <= 32 && (size_t)M * D * 2 < (1u << 23) && S < (1 << 23);
    if (head_major) {
        ALO_REQUIRE(wave_kernel && fused, ALO_ERR_UNSUPPORTED,
                    "alo_msda_forward_fused_hm: needs bf16, L = P = 4, D %% 8 == 0, D <= 32 and 16-byte aligned pointers");
        dm = make_dims(N, S, M, D, L, Lq, P, 16, 16384);
        dm.ref_dim = ref_dim;
        if (loc_row_elems > 0) dm.loc_row_elems = (int)loc_row_elems;
        if (attn_row_elems > 0) dm.attn_row_elems = (int)attn_row_elems;
        // runs are (16 consecutive queries, head) tiles; heads of one query block stay on neighbouring waves
        const long runs = (long)((Lq + 15) / 16) * M;
        dm.runs_per_batch = (int)runs;
        dm.blocks_per_batch = (int)((runs + dm.iters_per_block - 1) / dm.iters_per_block);
        dm.nblocks = (unsigned)(dm.blocks_per_batch * N);
        ResDims rd;
        const int rl = host_shapes && D == 32 ? resident_plan(host_shapes, N, S, M, L, Lq, resident_policy, &rd) : 0;
        const bool offs32 = (double)Lq * dm.loc_row_elems < 4.0e9 && (double)Lq * dm.attn_row_elems < 4.0e9 &&
                            (double)Lq * M * 32 < 4.0e9 && (double)Lq * 4 * ref_dim < 4.0e9;   // 32-bit element offsets per image
        if (rl && offs32) {
            // coarse levels resident in LDS (msda_fwd_bf16_resident_kernel): one 12-wave workgroup per CU pinned to an (image, head) slab
            dm.nblocks = (unsigned)((long)N * M * rd.wps);
            void* rargs[] = {&value, &spatial_shapes, &level_start_index, &loc, &attn, &ref, &out, &dm, &rd};
            const size_t lds = (size_t)rd.image_bytes + kResFixed + (size_t)kResWaves * kResWaveLds;
            static unsigned long long attr_done = 0;   // one bit per device
            const void* fn = reinterpret_cast<const void*>(msda_fwd_bf16_resident_kernel);
            hipError_t ea = ensure_dynamic_lds(fn, kResLdsTotal, &attr_done);
            if (ea != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_forward_fused_hm_resident: %s", hipGetErrorString(ea));
            hipError_t el = hipLaunchKernel(fn, dim3(dm.nblocks), dim3(kResThreads), rargs, lds, stream);
            if (el != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_forward_fused_hm_resident: %s", hipGetErrorString(el));
            return check_launch("alo_msda_forward_fused_hm_resident");
        }
        return launch(msda_fwd_bf16_mfma_kernel<4, true, true>, dm, kWaveLds, stream, "alo_msda_forward_fused_hm", args, 64);
    }
    if (wave_kernel) {
        // bf16 rows go to the matrix pipe untouched, one wave per 16 pairs (see msda_fwd_bf16_mfma_kernel)
        dm = make_dims(N, S, M, D, L, Lq, P, 16, 16384);
        dm.ref_dim = ref_dim;
        const char* what = fused ? "alo_msda_forward_fused" : "alo_msda_forward";
        if (fused) {
            return launch(msda_fwd_bf16_mfma_kernel<4, true, false>, dm, kWaveLds, stream, what, args, 64);
        }
        return launch(msda_fwd_bf16_mfma_kernel<4, false, false>, dm, kWaveLds, stream, what, args, 64);
    }
    if (value_dtype == ALO_F32) { ALO_ALL_CASES(ALO_FWD_CASE, float, float, float, 4) }
    if (value_dtype == ALO_F64) { ALO_ALL_CASES(ALO_FWD_CASE, double, double, double, 2) }
    if (value_dtype == ALO_BF16) { ALO_ALL_CASES(ALO_FWD_CASE, bf16_t, float, float, 8) }
    return fail(ALO_ERR_UNSUPPORTED, "alo_msda_forward: no kernel for vec=%d group=%d", plan.vec, plan.g);
}
}  // namespace

extern "C" int alo_msda_forward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                const void* sampling_loc, const void* attn_weight, void* out, int N, int S, int M,
                                int D, int L, int Lq, int P, int value_dtype, int loc_dtype, void* stream_) {
    return forward_impl(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, 0, out, N, S, M, D,
                        L, Lq, P, value_dtype, loc_dtype, stream_);
}

extern "C" int alo_msda_forward_fused(const void* value, const int32_t* spatial_shapes,
                                      const int32_t* level_start_index, const void* sampling_offsets,
                                      const void* attn_logits, const void* reference_points, void* out, int N, int S,
                                      int M, int D, int L, int Lq, int P, int ref_dim, int value_dtype, void* stream_) {
    ALO_REQUIRE(reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused: reference_points is null");
    ALO_REQUIRE(ref_dim == 2 || ref_dim == 4, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused: last dim of reference_points must be 2 or 4, got %d", ref_dim);
    // the geometry dtype is implied: fp64 for fp64 values, fp32 otherwise (loc_dtype only steers validation here)
    const int loc_dtype = value_dtype == ALO_F64 ? ALO_F64 : ALO_F32;
    return forward_impl(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                        ref_dim, out, N, S, M, D, L, Lq, P, value_dtype, loc_dtype, stream_);
}

extern "C" int alo_msda_forward_fused_hm(const void* value_hm, const int32_t* spatial_shapes,
                                         const int32_t* level_start_index, const void* sampling_offsets,
                                         const void* attn_logits, const void* reference_points, void* out, int N, int S,
                                         int M, int D, int L, int Lq, int P, int ref_dim, int value_dtype, void* stream_) {
    ALO_REQUIRE(reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused_hm: reference_points is null");
    ALO_REQUIRE(ref_dim == 2 || ref_dim == 4, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm: last dim of reference_points must be 2 or 4, got %d", ref_dim);
    return forward_impl(value_hm, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                        ref_dim, out, N, S, M, D, L, Lq, P, value_dtype, ALO_F32, stream_, true);
}

extern "C" int alo_msda_forward_fused_hm_rows(const void* value_hm, const int32_t* spatial_shapes,
                                              const int32_t* level_start_index, const void* sampling_offsets,
                                              const void* attn_logits, long offsets_row_elems, long logits_row_elems,
                                              const void* reference_points, void* out, int N, int S, int M, int D, int L,
                                              int Lq, int P, int ref_dim, int value_dtype, void* stream_) {
    ALO_REQUIRE(reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused_hm_rows: reference_points is null");
    ALO_REQUIRE(ref_dim == 2 || ref_dim == 4, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm_rows: last dim of reference_points must be 2 or 4, got %d", ref_dim);
    ALO_REQUIRE(offsets_row_elems >= (long)M * L * P * 2 && logits_row_elems >= (long)M * L * P && offsets_row_elems % 8 == 0 &&
                    logits_row_elems % 8 == 0 && offsets_row_elems < (1L << 30) && logits_row_elems < (1L << 30),
                ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm_rows: row strides must cover a query's M*L*P*2 offsets / M*L*P logits and keep 16-byte alignment");
    return forward_impl(value_hm, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                        ref_dim, out, N, S, M, D, L, Lq, P, value_dtype, ALO_F32, stream_, true, offsets_row_elems,
                        logits_row_elems);
}

extern "C" int alo_msda_forward_fused_hm_resident(const void* value_hm, const int32_t* spatial_shapes,
                                                  const int32_t* level_start_index, const void* sampling_offsets,
                                                  const void* attn_logits, long offsets_row_elems, long logits_row_elems,
                                                  const void* reference_points, void* out, int N, int S, int M, int D, int L,
                                                  int Lq, int P, int ref_dim, int value_dtype, const int32_t* host_spatial_shapes,
                                                  int policy, void* stream_) {
    ALO_REQUIRE(reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused_hm_resident: reference_points is null");
    ALO_REQUIRE(host_spatial_shapes, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused_hm_resident: host_spatial_shapes is null");
    ALO_REQUIRE(policy == ALO_RESIDENT_AUTO || policy == ALO_RESIDENT_ALWAYS, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm_resident: policy must be ALO_RESIDENT_AUTO or ALO_RESIDENT_ALWAYS, got %d", policy);
    ALO_REQUIRE(ref_dim == 2 || ref_dim == 4, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm_resident: last dim of reference_points must be 2 or 4, got %d", ref_dim);
    ALO_REQUIRE(offsets_row_elems >= (long)M * L * P * 2 && logits_row_elems >= (long)M * L * P && offsets_row_elems % 8 == 0 &&
                    logits_row_elems % 8 == 0 && offsets_row_elems < (1L << 30) && logits_row_elems < (1L << 30),
                ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused_hm_resident: row strides must cover a query's M*L*P*2 offsets / M*L*P logits and keep 16-byte alignment");
    return forward_impl(value_hm, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                        ref_dim, out, N, S, M, D, L, Lq, P, value_dtype, ALO_F32, stream_, true, offsets_row_elems,
                        logits_row_elems, host_spatial_shapes, policy);
}

extern "C" int alo_msda_resident_levels(const int32_t* host_spatial_shapes, int N, int S, int M, int L, int Lq, int policy) {
    ResDims rd;
    return host_spatial_shapes ? resident_plan(host_spatial_shapes, N, S, M, L, Lq, policy, &rd) : 0;
}

namespace {
// ALO_MSDA_BWD = "tiled" keeps msda_bwd_tiled_kernel for the encoder-shaped launches the wide path would take (measurement knob)
int bwd_policy() {   // read per call: a test flips it inside one process
    const char* e = getenv("ALO_MSDA_BWD");
    return (e && !strcmp(e, "tiled")) ? 1 : 0;
}

int backward_impl(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                  const void* sampling_loc, const void* attn_weight, const void* grad_out, void* grad_value,
                  void* grad_sampling_loc, void* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                  int value_dtype, int loc_dtype, const int32_t* host_shapes, void* stream_) {
    size_t elem = 0;
    if (int rc = validate(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P,
                          value_dtype, loc_dtype, &elem))
        return rc;
    ALO_REQUIRE(grad_out && grad_value && grad_sampling_loc && grad_attn_weight, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_backward: null gradient pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t gelem = value_dtype == ALO_F64 ? 8 : 4;
    hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * gelem, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_backward: memset: %s", hipGetErrorString(e));
    const bool aligned = (((uintptr_t)value | (uintptr_t)grad_out) & 15) == 0;
    const bool all_aligned = aligned && (((uintptr_t)sampling_loc | (uintptr_t)attn_weight | (uintptr_t)grad_value |
                                          (uintptr_t)grad_sampling_loc | (uintptr_t)grad_attn_weight) & 15) == 0;
    const bool frame32 = (double)S * M * 128 < 4.0e9 && (double)Lq * M * 32 < 2.0e9;   // 32-bit byte offsets inside one frame
    if ((value_dtype == ALO_F32 || value_dtype == ALO_BF16) && (D == 32 || D == 64) && L == 4 && P == 4 && all_aligned && frame32 && host_shapes &&
        Lq == S && bwd_policy() != 1) {
        // queries = the pyramid's own pixels: 16x16 query blocks, sorted on chip (msda_bwd_wide.hip)
        const int rc = msda_backward_wide(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                                          grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, value_dtype, host_shapes, stream);
        if (rc != ALO_ERR_UNSUPPORTED) return rc;
    }
    if (value_dtype == ALO_F32 && D == 32 && L == 4 && P == 4 && all_aligned && frame32) {
        // the DETR-family shape: tiled, window-dense backward on the fp32 matrix cores (msda_bwd_tiled_kernel)
        TileDims td;
        td.S = S; td.M = M; td.Lq = Lq;
        td.pyramid = 0;
        td.n_super = (Lq + 15) / 16;
        const int tile = 4;
        if (host_shapes && Lq == S) {
            // queries = the pyramid's own pixels (encoder self-attention): 4x4 blocks of each level.  The host copy of the shapes
            // only SIZES the grid; the kernel derives which queries a tile owns from the device copy.  A host copy that disagrees
            // with the device one can therefore only leave tiles empty or — if it under-counts — miss queries, which is why the
            // Python host never caches it across tensors (alo_hip.msda_backward).
            long total = 0;
            int tiles = 0;
            bool ok = true;
            for (int l = 0; l < 4; ++l) {
                const int h = host_shapes[2 * l], w = host_shapes[2 * l + 1];
                ok = ok && h > 0 && w > 0;
                tiles += ((h + tile - 1) / tile) * ((w + tile - 1) / tile);
                total += (long)h * w;
            }
            if (ok && total == S) { td.pyramid = 1; td.n_super = tiles; }
        }
        const long nb = (long)N * td.n_super * M;
        ALO_REQUIRE(nb < 0x7fffffffL, ALO_ERR_UNSUPPORTED, "alo_msda_backward: grid too large");
        td.nblocks = (unsigned)nb;
        {
            static unsigned long long attr_done = 0;   // one bit per device
            hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void*>(msda_bwd_tiled_kernel), kTileLds, &attr_done);
            if (ea != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_backward: %s", hipGetErrorString(ea));
        }
        void* targs[] = {&value, &spatial_shapes, &level_start_index, &sampling_loc, &attn_weight, &grad_out,
                         &grad_value, &grad_sampling_loc, &grad_attn_weight, &td};
        hipError_t el = hipLaunchKernel(reinterpret_cast<const void*>(msda_bwd_tiled_kernel), dim3(td.nblocks), dim3(64),
                                        targs, kTileLds, stream);
        if (el != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_backward: %s", hipGetErrorString(el));
        return check_launch("alo_msda_backward");
    }
    Plan plan = make_plan(D, L, P, elem, aligned);
    if (D <= 64) {
        // one channel per lane: each of the four atomics of a sampling point then covers D consecutive elements of ONE row
        // (a whole 128-byte line for D = 32 fp32) instead of every VEC-th element of it
        plan.vec = 1;
        plan.g = D <= 8 ? 8 : (D <= 32 ? 32 : 64);
        plan.lp16 = false;
    }
    Dims dm = make_dims(N, S, M, D, L, Lq, P, plan.g);
    void* args[] = {&value, &spatial_shapes, &level_start_index, &sampling_loc, &attn_weight, &grad_out,
                    &grad_value, &grad_sampling_loc, &grad_attn_weight, &dm};
    if (value_dtype == ALO_F32) { ALO_ALL_CASES(ALO_BWD_CASE, float, float, float, 4) }
    if (value_dtype == ALO_F64) { ALO_ALL_CASES(ALO_BWD_CASE, double, double, double, 2) }
    if (value_dtype == ALO_BF16) { ALO_ALL_CASES(ALO_BWD_CASE, bf16_t, float, float, 8) }
    return fail(ALO_ERR_UNSUPPORTED, "alo_msda_backward: no kernel for vec=%d group=%d", plan.vec, plan.g);
}
}  // namespace

extern "C" int alo_msda_backward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                 const void* sampling_loc, const void* attn_weight, const void* grad_out,
                                 void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int N, int S, int M,
                                 int D, int L, int Lq, int P, int value_dtype, int loc_dtype, void* stream_) {
    return backward_impl(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                         grad_sampling_loc, grad_attn_weight, N, S, M, D, L, Lq, P, value_dtype, loc_dtype, nullptr, stream_);
}

extern "C" int alo_msda_backward_hinted(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                        const void* sampling_loc, const void* attn_weight, const void* grad_out,
                                        void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int N, int S,
                                        int M, int D, int L, int Lq, int P, int value_dtype, int loc_dtype,
                                        const int32_t* host_spatial_shapes, void* stream_) {
    return backward_impl(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, grad_value,
                         grad_sampling_loc, grad_attn_weight, N, S, M, D, L, Lq, P, value_dtype, loc_dtype,
                         host_spatial_shapes, stream_);
}

// Which kernel alo_msda_backward[_hinted] takes for a launch of these dimensions (16-byte aligned pointers assumed): the answer the
// dispatch in backward_impl gives, without enqueuing anything.
extern "C" int alo_msda_backward_path(int N, int S, int M, int D, int L, int Lq, int P, int value_dtype, int loc_dtype,
                                      const int32_t* host_spatial_shapes) {
    const bool pair_ok = (value_dtype == ALO_F32 && loc_dtype == ALO_F32) || (value_dtype == ALO_F64 && loc_dtype == ALO_F64) ||
                         (value_dtype == ALO_BF16 && loc_dtype == ALO_F32);
    if (!pair_ok || N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return -1;
    const bool frame32 = (double)S * M * 128 < 4.0e9 && (double)Lq * M * 32 < 2.0e9;
    if ((value_dtype == ALO_F32 || value_dtype == ALO_BF16) && (D == 32 || D == 64) && L == 4 && P == 4 && frame32 && host_spatial_shapes && Lq == S &&
        bwd_policy() != 1 &&
        msda_backward_wide(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, S, M, D, Lq, value_dtype,
                           host_spatial_shapes, nullptr, true) == ALO_OK)
        return ALO_MSDA_BWD_WIDE;
    if (value_dtype == ALO_F32 && D == 32 && L == 4 && P == 4 && frame32) return ALO_MSDA_BWD_TILED;
    return ALO_MSDA_BWD_GENERIC;
}
