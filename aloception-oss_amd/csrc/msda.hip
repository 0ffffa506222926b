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
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace alo {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxLevels = 32;
constexpr int kMetaBytes = 512;  // H[32] | W[32] | start[32] as int32, padded

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
    unsigned nblocks;
    int ref_dim;           // fused prologue only: last dim of reference_points (2 or 4)
    int ablate;            // tuning experiments only (ALO_MSDA_ABLATE): 1 = no gathers, 2 = descriptors built once
};

// Image-space position, validity and the four corners of one sampling point (cuh:285-291, :38-78).
template <typename CT>
struct Tap {
    bool valid, ok[4];
    int base;  // pixel index (within the batch item's S rows) of the (h_low, w_low) corner
    int W;
    int h_low, w_low;
    CT lh, lw;
};
template <typename CT>
__device__ __forceinline__ Tap<CT> make_tap(CT loc_x, CT loc_y, int H, int W, int start) {
    Tap<CT> t;
    const CT h_im = loc_y * (CT)H - (CT)0.5;
    const CT w_im = loc_x * (CT)W - (CT)0.5;
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
    t.base = start + h_low * W + w_low;
    t.W = W;
    t.h_low = h_low;
    t.w_low = w_low;
    return t;
}

__device__ __forceinline__ void load_meta(int* meta, const int32_t* shapes, const int32_t* lstart, int L) {
    const int t = threadIdx.x;
    if (t < L) {
        meta[t] = shapes[2 * t];
        meta[kMaxLevels + t] = shapes[2 * t + 1];
        meta[2 * kMaxLevels + t] = lstart[t];
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
        const int nsamp = (dm.ablate == 2 && it > 0) ? 0 : PAIRS * LP;
        for (int si = tid; si < nsamp; si += kThreads) {
            const int pl = si / LP, s = si - pl * LP;
            const int pair = pair0 + pl;
            Desc d;
#pragma unroll
            for (int k = 0; k < 4; ++k) { d.off[k] = kOutOfRange; d.w[k] = (CT)0; }
            const bool live = pair < dm.pairs_per_batch;
            const long g = (batch_pair0 + (live ? pair : 0)) * LP + s;
            CT x = (CT)0, y = (CT)0, a = (CT)0;
            const int l = s / dm.P;
            if (live) { x = (CT)ld(loc + 2 * g); y = (CT)ld(loc + 2 * g + 1); a = (CT)ld(attn + g); }
            if constexpr (FUSED) {
                // softmax over the pair's L*P logits
                CT mx, sum;
                if constexpr (LP_CT == 16) {  // the 16 samples of a pair sit in 16 consecutive, aligned lanes
                    mx = a;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                    a = exp(a - mx);
                    sum = a;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                } else {
                    const long g0 = g - s;
                    mx = (CT)ld(attn + g0);
                    for (int j = 1; j < LP; ++j) mx = fmax(mx, (CT)ld(attn + g0 + j));
                    sum = (CT)0;
                    for (int j = 0; j < LP; ++j) sum += exp((CT)ld(attn + g0 + j) - mx);
                    a = exp(a - mx);
                }
                a = a / sum;
                if (live) {
                    const int q = pair / dm.M;
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
            if (live) {
                const Tap<CT> t = make_tap<CT>(x, y, meta[l], meta[kMaxLevels + l], meta[2 * kMaxLevels + l]);
                const CT hh = (CT)1 - t.lh, hw = (CT)1 - t.lw;
                const CT w[4] = {hh * hw, hh * t.lw, t.lh * hw, t.lh * t.lw};
                const int px[4] = {t.base, t.base + 1, t.base + t.W, t.base + t.W + 1};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (t.ok[k]) { d.off[k] = (unsigned)px[k] * row_bytes; d.w[k] = w[k] * a; }
                }
            }
            *reinterpret_cast<Desc*>(dbase + pl * pair_stride + s * (int)sizeof(Desc)) = d;
        }
        __syncthreads();

        // ---- stage 2: gather.  G lanes cover the channels of one pair ----------------------------------------------
        {
            const int pl = tid / G, lane = tid % G;
            const int pair = pair0 + pl;
            if (pair < dm.pairs_per_batch) {
                const int m = pair % dm.M;
                const unsigned char* dp = dbase + pl * pair_stride;
                for (int c0 = lane * VEC; c0 < dm.D; c0 += G * VEC) {
                    const unsigned coff = (unsigned)(m * dm.D + c0) * (unsigned)sizeof(T);
                    CT acc[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = (CT)0;
                    // SB sampling points (4*SB corner rows) are requested before the first one is consumed; the
                    // outer loop stays rolled so the register allocator sees exactly that much in flight.
#pragma unroll 1
                    for (int s0 = 0; s0 < LP; s0 += SB) {
                        Desc d[SB];
                        typename Ld::raw_t raw[SB][4];
#pragma unroll
                        for (int j = 0; j < SB; ++j) {
                            if (LP_CT || s0 + j < LP) {
                                d[j] = *reinterpret_cast<const Desc*>(dp + (s0 + j) * (int)sizeof(Desc));
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k) { d[j].off[k] = kOutOfRange; d[j].w[k] = (CT)0; }
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                raw[j][k] = Ld::load(rsrc, dm.ablate == 1 ? kOutOfRange : d[j].off[k] + coff);
                        }
#pragma unroll
                        for (int j = 0; j < SB; ++j) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                CT v[VEC];
                                Ld::widen(raw[j][k], v);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[i] += d[j].w[k] * v[i];
                            }
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
// forward, tiled encoder variant: sample footprint staged in LDS
// ------------------------------------------------------------------------------------------------------------------
// The generic kernel above is bound by the rate at which a CU's texture path accepts gather lanes (64 B/clk): every
// (query, head, point) costs four 64/128-byte requests whether or not they hit in L1 (measured: with the gathers pointed
// out of range the kernel still takes 80 % of its time).  In the encoder the queries ARE the pixels of the pyramid, and
// neighbouring queries sample overlapping neighbourhoods.  Here one workgroup owns (image, head, TY x 8 tile of query
// pixels of one level): stage 1 builds the tile's sampling taps and, per value level, the bounding box of every corner
// they touch; the boxes (one head's D channels per pixel) are copied into LDS once; the gather then runs out of LDS at
// ds_read_b128 rate.  A level whose box does not fit the LDS budget is gathered from global memory exactly as in the
// generic kernel, so nothing is assumed about the sampling locations — a wider spread only costs speed.
constexpr int kTileLevels = 4;
struct TileArgs {
    const void* value; const void* loc; const void* attn; const void* ref; void* out;
    int N, S, M, D, Lq, ref_dim;
    int H[kTileLevels], W[kTileLevels], start[kTileLevels];
    int tiles_x[kTileLevels], tile_prefix[kTileLevels + 1];
    int tiles_total;
    unsigned nblocks;
    int window_budget;  // bytes of LDS available for staged windows
};

// arr[i] for a runtime i without dynamically indexing a kernel-argument array (which would send the whole argument
// struct to scratch memory): a chain of selects over compile-time indices.
template <int N>
__device__ __forceinline__ int pick(const int (&arr)[N], int i) {
    int v = arr[0];
#pragma unroll
    for (int l = 1; l < N; ++l) v = (i == l) ? arr[l] : v;
    return v;
}

// Value of `v` in lane (lane ^ MASK): one DPP quad_perm for MASK < 4, a cross-lane shuffle otherwise.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
    if constexpr (MASK == 1) {
        return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xf, 0xf, true));  // [1,0,3,2]
    } else if constexpr (MASK == 2) {
        return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xf, 0xf, true));  // [2,3,0,1]
    } else {
        return __shfl_xor(v, MASK, 64);
    }
}

// Reduce-scatter inside an aligned group of G lanes: every lane holds partial sums for all G*VEC channels of its query
// (acc[g][i] = channel g*VEC + i); on return lane j's acc[j][*] holds the group total of "its" VEC channels.
template <int G, int VEC>
__device__ __forceinline__ void group_reduce_scatter(float (&acc)[G][VEC], int part) {
#pragma unroll
    for (int half = G / 2; half >= 1; half /= 2) {
        // lanes whose `half` bit is 0 keep the lower block of `half` chunks of the current range, the others the upper
        const bool upper = (part & half) != 0;
#pragma unroll
        for (int gidx = 0; gidx < half; ++gidx)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                // current range is [base, base + 2*half) with base determined by the bits above `half`; since all
                // indices must be compile-time we process every aligned block of 2*half chunks identically
#pragma unroll
                for (int base = 0; base < G; base += 2 * half) {
                    const float lo = acc[base + gidx][i], hi = acc[base + half + gidx][i];
                    const float send = upper ? lo : hi;   // what the partner keeps
                    float got;
                    if (half == 1) got = lane_xor<1>(send);
                    else if (half == 2) got = lane_xor<2>(send);
                    else got = lane_xor<4>(send);
                    const float keep = upper ? hi : lo;
                    // result parked in the slot this lane keeps
                    if (upper) acc[base + half + gidx][i] = keep + got; else acc[base + gidx][i] = keep + got;
                }
            }
    }
}

template <typename T, typename LT, typename CT, int VEC, int G, bool FUSED>
__global__ void __launch_bounds__(kThreads, 4)
msda_fwd_tile_kernel(const TileArgs a) {
    static_assert(sizeof(CT) == 4, "tiled kernel computes in fp32");
    constexpr int TQ = kThreads / G;  // queries per tile
    constexpr int TX = 8, TY = TQ / TX;
    constexpr int L = kTileLevels, P = 4, LP = L * P;
    constexpr int ROWB = G * 16;       // bytes of one (pixel, head) row: D * sizeof(T)
    constexpr int NT = LP / G;         // sampling points owned by each lane of a query's group (4 or 2)
    constexpr int RPB = 16 / G;        // pixel rows per 256-byte LDS bank row
    using Ld = Loader<T, CT, VEC>;
    using InT = typename std::conditional<FUSED, T, LT>::type;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* bbox = reinterpret_cast<int*>(smem);  // [L][ymin, ymax, xmin, xmax]
    unsigned char* win = smem + 64;            // [zero row (ROWB bytes)] [staged windows]

    const InT* loc = static_cast<const InT*>(a.loc);
    const InT* attn = static_cast<const InT*>(a.attn);
    const CT* ref = static_cast<const CT*>(a.ref);
    const T* value = static_cast<const T*>(a.value);
    T* out = static_cast<T*>(a.out);

    const unsigned lb = xcd_contiguous_block(blockIdx.x, a.nblocks);
    const int m = lb % a.M;
    const int t = (lb / a.M) % a.tiles_total;
    const int b = lb / (a.M * a.tiles_total);
    int lq = 0;
#pragma unroll
    for (int l = 1; l < L; ++l)
        if (t >= a.tile_prefix[l]) lq = l;
    const int tt = t - pick(a.tile_prefix, lq);
    const int tiles_x = pick(a.tiles_x, lq);
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int Hq = pick(a.H, lq), Wq = pick(a.W, lq), startq = pick(a.start, lq);
    const int tid = threadIdx.x;

    if (tid < 4 * L) bbox[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    if (tid < ROWB / 4) reinterpret_cast<unsigned*>(win)[tid] = 0u;
    __syncthreads();

    const unsigned row_elems = (unsigned)a.M * a.D;
    const unsigned row_bytes = row_elems * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(value + (size_t)b * a.S * row_elems, (unsigned)a.S * row_bytes);

    // ---- pass A: lane `part` of a query's group looks at the NT consecutive sampling points it owns: softmax statistics
    //      of the query's 16 logits (fused variant) and the bounding box of the corners touched, per value level ----------
    const int ql = tid / G, part = tid % G;
    const int qy = ty * TY + ql / TX, qx = tx * TX + ql % TX;
    const bool live = qy < Hq && qx < Wq;
    const int q = startq + (live ? qy * Wq + qx : 0);
    const int s0 = part * NT;
    const int lv = s0 / P;  // G = 4: lane j owns level j; G = 8: lanes 2l, 2l+1 share level l
    const int Hl = pick(a.H, lv), Wl = pick(a.W, lv), startl = pick(a.start, lv);
    const long g0 = (((long)b * a.Lq + q) * a.M + m) * LP + s0;

    CT r0 = (CT)0, r1 = (CT)0, r2 = (CT)0, r3 = (CT)0;  // reference point of (query, level) — fused variant only
    CT smax = (CT)0, sinv = (CT)1;                      // softmax: max logit and 1 / sum(exp)
    // raw inputs of the lane's NT points stay in registers for both passes
    CT rx[NT], ry[NT], rw[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        rx[k] = live ? (CT)ld(loc + 2 * (g0 + k)) : (CT)0;
        ry[k] = live ? (CT)ld(loc + 2 * (g0 + k) + 1) : (CT)0;
        rw[k] = live ? (CT)ld(attn + g0 + k) : (CT)0;
    }
    if constexpr (FUSED) {
        if (live) {
            const CT* r = ref + (((long)b * a.Lq + q) * L + lv) * a.ref_dim;
            r0 = r[0]; r1 = r[1];
            if (a.ref_dim == 4) { r2 = r[2]; r3 = r[3]; }
        }
        smax = rw[0];
#pragma unroll
        for (int k = 1; k < NT; ++k) smax = fmax(smax, rw[k]);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o, 64));
        CT sum = (CT)0;
#pragma unroll
        for (int k = 0; k < NT; ++k) sum += exp(rw[k] - smax);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        sinv = (CT)1 / sum;
    }
    // sampling location and attention weight of a point from its raw inputs, as the generic kernel's stage 1 computes them
    auto point = [&](CT x, CT y, CT w, CT& ox, CT& oy, CT& ow) {
        if constexpr (FUSED) {
            ow = exp(w - smax) * sinv;
            if (a.ref_dim == 2) {
                ox = r0 + x / (CT)Wl;
                oy = r1 + y / (CT)Hl;
            } else {
                ox = r0 + x / (CT)P * r2 * (CT)0.5;
                oy = r1 + y / (CT)P * r3 * (CT)0.5;
            }
        } else {
            ox = x; oy = y; ow = w;
        }
    };
    int ylo = INT_MAX, yhi = INT_MIN, xlo = INT_MAX, xhi = INT_MIN;
    if (live) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            CT x, y, w;
            point(rx[k], ry[k], rw[k], x, y, w);
            const Tap<CT> tp = make_tap<CT>(x, y, Hl, Wl, 0);
            if (tp.valid) {  // rows / columns of the map this point really touches
                ylo = min(ylo, max(tp.h_low, 0));
                yhi = max(yhi, min(tp.h_low + 1, Hl - 1));
                xlo = min(xlo, max(tp.w_low, 0));
                xhi = max(xhi, min(tp.w_low + 1, Wl - 1));
            }
        }
    }
    // lanes of the same level differ in every lane bit except the ones that select the level
    constexpr int LVL_LANES = G / L;  // lanes per level inside a group (1 for G = 4, 2 for G = 8)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        if (o >= LVL_LANES && o < G) continue;
        ylo = min(ylo, __shfl_xor(ylo, o, 64));
        yhi = max(yhi, __shfl_xor(yhi, o, 64));
        xlo = min(xlo, __shfl_xor(xlo, o, 64));
        xhi = max(xhi, __shfl_xor(xhi, o, 64));
    }
    if ((tid & 63) < G && (tid & (LVL_LANES - 1)) == 0) {  // one lane per level per wave
        atomicMin(&bbox[lv * 4 + 0], ylo);
        atomicMax(&bbox[lv * 4 + 1], yhi);
        atomicMin(&bbox[lv * 4 + 2], xlo);
        atomicMax(&bbox[lv * 4 + 3], xhi);
    }
    __syncthreads();

    // ---- window plan: identical in every thread.  Either all four boxes fit the LDS budget (the normal case) or the whole
    //      tile gathers from global memory — a workgroup-uniform decision keeps both gather loops branch-free ------------------
    int my_y0 = 0, my_x0 = 0, my_ww = 0, my_base = 0;
    int used = ROWB;  // the zero row comes first
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int y0 = bbox[l * 4 + 0], y1 = bbox[l * 4 + 1], x0 = bbox[l * 4 + 2], x1 = bbox[l * 4 + 3];
        const bool any = y0 <= y1 && x0 <= x1;
        const int h = any ? y1 - y0 + 1 : 0, w = any ? x1 - x0 + 1 : 0;
        if (l == lv) { my_y0 = any ? y0 : 0; my_x0 = any ? x0 : 0; my_ww = w; my_base = used; }
        const long nb = (long)h * w * ROWB;
        used = (nb > (long)a.window_budget) ? a.window_budget + 1 : used + (int)nb;  // saturate instead of overflowing
    }
    const bool in_lds = used <= a.window_budget;

    if (in_lds) {
        // copy the boxes: one head's D channels of every pixel.  All levels form ONE flat list of 16-byte chunks (their
        // positions in LDS); a thread puts up to 8 loads in flight before the first ds_write.  The G chunks of a pixel are
        // XOR-swizzled so that 16 lanes reading the same chunk index of 16 consecutive pixels hit 16 different bank slots.
        int lb0[L], ly0[L], lx0[L], lw[L];
        {
            int base = ROWB;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int y0 = bbox[l * 4 + 0], y1 = bbox[l * 4 + 1], x0 = bbox[l * 4 + 2], x1 = bbox[l * 4 + 3];
                const bool any = y0 <= y1 && x0 <= x1;
                lb0[l] = base; ly0[l] = any ? y0 : 0; lx0[l] = any ? x0 : 0; lw[l] = any ? x1 - x0 + 1 : 1;
                base += any ? (y1 - y0 + 1) * (x1 - x0 + 1) * ROWB : 0;
            }
        }
        const int total = (used - ROWB) / 16;
        constexpr int NF = 8;
        for (int i0 = tid; i0 < total; i0 += NF * kThreads) {
            u32x4 v[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int i = i0 + j * kThreads;
                const int pos = ROWB + i * 16;  // byte position of the chunk in the window area
                int l = 0;
#pragma unroll
                for (int t2 = 1; t2 < L; ++t2)
                    if (pos >= lb0[t2]) l = t2;
                int base = lb0[0], y0 = ly0[0], x0 = lx0[0], w = lw[0];
#pragma unroll
                for (int t2 = 1; t2 < L; ++t2)
                    if (l == t2) { base = lb0[t2]; y0 = ly0[t2]; x0 = lx0[t2]; w = lw[t2]; }
                const int rel = pos - base;
                const int r = rel / ROWB, cs = (rel / 16) & (G - 1);
                const int c = cs ^ ((r / RPB) & (G - 1));
                int yy = (int)(((float)r + 0.5f) / (float)w);
                int xx = r - yy * w;
                if (xx < 0) { --yy; xx += w; }
                if (xx >= w) { ++yy; xx -= w; }
                const unsigned goff = (unsigned)(pick(a.start, l) + (y0 + yy) * pick(a.W, l) + x0 + xx) * row_bytes +
                                      (unsigned)(m * a.D) * (unsigned)sizeof(T) + (unsigned)c * 16u;
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i < total ? goff : kOutOfRange, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int i = i0 + j * kThreads;
                if (i < total) *reinterpret_cast<u32x4*>(win + ROWB + i * 16) = v[j];
            }
        }
        __syncthreads();
    }

    // ---- pass B: walk the lane's points again (inputs are L1-hot), gather each corner row over all G*VEC channels -------------
    CT acc[G][VEC];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[gi][i] = (CT)0;
    const unsigned head_off = (unsigned)(m * a.D) * (unsigned)sizeof(T);
    if (live) {
        // rolled on purpose (an unrolled body makes the scheduler hoist every read and spill); the point inputs rotate through
        // slot 0 so that no register array is indexed dynamically
#pragma unroll 1
        for (int k = 0; k < NT; ++k) {
            CT x, y, w;
            point(rx[0], ry[0], rw[0], x, y, w);
#pragma unroll
            for (int j = 0; j + 1 < NT; ++j) { rx[j] = rx[j + 1]; ry[j] = ry[j + 1]; rw[j] = rw[j + 1]; }
            const Tap<CT> tp = make_tap<CT>(x, y, Hl, Wl, 0);
            const CT hh = (CT)1 - tp.lh, hw = (CT)1 - tp.lw;
            const CT wk[4] = {hh * hw * w, hh * tp.lw * w, tp.lh * hw * w, tp.lh * tp.lw * w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = tp.h_low + (c >> 1), xx = tp.w_low + (c & 1);
                typename Ld::raw_t raw[G];
                if (in_lds) {
                    const int r = (yy - my_y0) * my_ww + (xx - my_x0);
                    const unsigned rowo = tp.ok[c] ? (unsigned)(my_base + r * ROWB) : 0u;  // 0 = the zero row
                    const unsigned key = tp.ok[c] ? (unsigned)((r / RPB) & (G - 1)) : 0u;
#pragma unroll
                    for (int gi = 0; gi < G; ++gi)
                        raw[gi] = *reinterpret_cast<const typename Ld::raw_t*>(win + rowo + ((gi ^ key) << 4));
                } else {
                    const unsigned o = (unsigned)(startl + yy * Wl + xx) * row_bytes + head_off;
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) raw[gi] = Ld::load(rsrc, tp.ok[c] ? o + gi * 16 : kOutOfRange);
                }
                const CT wc = tp.ok[c] ? wk[c] : (CT)0;
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    CT v[VEC];
                    Ld::widen(raw[gi], v);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[gi][i] += wc * v[i];
                }
            }
        }
    }

    // ---- sum the group's partial results; lane j ends up with channels [j*VEC, (j+1)*VEC) -------------------------------------
    group_reduce_scatter<G, VEC>(acc, part);
    CT mine[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        CT v = acc[0][i];
#pragma unroll
        for (int gi = 1; gi < G; ++gi) v = (part == gi) ? acc[gi][i] : v;
        mine[i] = v;
    }
    if (live) store_vec<T, CT, VEC>(out + (((long)b * a.Lq + q) * a.M + m) * a.D + part * VEC, mine);
}

// Can the tiled kernel take this call?  (host copy of the shapes known, the model's L = P = 4, queries = pixels,
// channels of a head = one 16-byte load per lane for 4 or 8 lanes)
bool tile_plan(TileArgs& ta, const int32_t* shapes_host, int N, int S, int M, int D, int L, int Lq, int P, size_t elem,
               bool aligned) {
    if (!shapes_host || L != kTileLevels || P != 4 || Lq != S || !aligned) return false;
    const int vec = (int)(16 / elem);
    if (D % vec != 0) return false;
    const int g = D / vec;
    if (!(elem == 2 && g == 4)) return false;  // bf16, D = 32 (the fp32 instantiation exists but still spills registers)
    const int tq = kThreads / g, tyq = tq / 8;
    long total = 0, start = 0;
    for (int l = 0; l < L; ++l) {
        const int h = shapes_host[2 * l], w = shapes_host[2 * l + 1];
        if (h <= 0 || w <= 0) return false;
        ta.H[l] = h; ta.W[l] = w; ta.start[l] = (int)start;
        ta.tiles_x[l] = (w + 7) / 8;
        ta.tile_prefix[l] = (int)total;
        total += (long)ta.tiles_x[l] * ((h + tyq - 1) / tyq);
        start += (long)h * w;
    }
    if (start != S) return false;
    ta.tile_prefix[L] = (int)total;
    ta.tiles_total = (int)total;
    const long nblocks = total * M * N;
    if (nblocks <= 0 || nblocks >= 0x7fffffffL) return false;
    ta.nblocks = (unsigned)nblocks;
    return true;
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
#pragma unroll
                for (int off = G / 2; off > 0; off >>= 1) {
                    s_attn += __shfl_xor(s_attn, off, 64);
                    s_w += __shfl_xor(s_w, off, 64);
                    s_h += __shfl_xor(s_h, off, 64);
                }
                if (live && lane == 0) {
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
// host dispatch
// ------------------------------------------------------------------------------------------------------------------
struct Plan {
    int vec, g;
    bool lp16;
};

// Kernel-tuning knobs, read once from the environment (results never depend on them).
//   ALO_MSDA_FWD_BATCH  2 | 4 | 8   sampling points whose 4 corner loads are put in flight together (forward)
//   ALO_MSDA_ITERS      1..64       runs of pairs per workgroup (0 = automatic)
//   ALO_MSDA_TILE       0 | 1       tiled (LDS-staged) encoder kernel off / on;  ALO_MSDA_TILE_LDS = its window budget
struct Tuning {
    int fwd_batch = 4;
    int iters = 0;
    int ablate = 0;
    int tile = 1;          // ALO_MSDA_TILE=0 disables the tiled encoder kernel
    int tile_lds = 36 * 1024;  // ALO_MSDA_TILE_LDS: bytes of LDS for staged windows (4 workgroups / CU at 36 KB)
};
const Tuning& tuning() {
    static const Tuning t = [] {
        Tuning x;
        if (const char* e = getenv("ALO_MSDA_FWD_BATCH")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) x.fwd_batch = v; }
        if (const char* e = getenv("ALO_MSDA_ABLATE")) x.ablate = atoi(e);
        if (const char* e = getenv("ALO_MSDA_TILE")) x.tile = atoi(e);
        if (const char* e = getenv("ALO_MSDA_TILE_LDS")) { const int v = atoi(e); if (v >= 0 && v <= 120 * 1024) x.tile_lds = v; }
        if (const char* e = getenv("ALO_MSDA_ITERS")) { const int v = atoi(e); if (v >= 0 && v <= 64) x.iters = v; }
        return x;
    }();
    return t;
}

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

Dims make_dims(int N, int S, int M, int D, int L, int Lq, int P, int G) {
    Dims d;
    d.S = S; d.M = M; d.D = D; d.L = L; d.P = P;
    d.pairs_per_batch = Lq * M;
    d.ref_dim = 0;
    d.ablate = tuning().ablate;
    const int pairs = kThreads / G;
    const long iters_total = ((long)d.pairs_per_batch + pairs - 1) / pairs;
    long ipb = iters_total * N / 4096;  // keep >= ~4096 workgroups in flight when the problem allows it
    if (ipb < 1) ipb = 1;
    if (ipb > 8) ipb = 8;
    if (tuning().iters > 0) ipb = tuning().iters;
    d.iters_per_block = (int)ipb;
    d.blocks_per_batch = (int)((iters_total + ipb - 1) / ipb);
    d.nblocks = (unsigned)(d.blocks_per_batch * N);
    return d;
}

template <typename K>
int launch(K kernel, const Dims& dm, size_t lds, hipStream_t stream, const char* what, void** args) {
    if (lds > 160 * 1024) return fail(ALO_ERR_UNSUPPORTED, "%s: L*P too large for LDS (%zu bytes)", what, lds);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    }
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), dim3(dm.nblocks), dim3(kThreads), args, lds,
                                   stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return check_launch(what);
}

#define ALO_FWD_CASE(T, LT, CT, VEC, G, LPCT)                                                                     \
    if (plan.vec == VEC && plan.g == G && plan.lp16 == (LPCT == 16)) {                                             \
        const size_t lds = kMetaBytes + (size_t)(kThreads / G) * ((size_t)L * P * sizeof(FwdDesc<CT>) + 16);       \
        if (fused) {                                                                                               \
            if (LPCT == 16 && sb == 2)                                                                             \
                return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, 2, true>, dm, lds, stream, "alo_msda_forward_fused", args); \
            return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, (LPCT ? 4 : 2), true>, dm, lds, stream, "alo_msda_forward_fused", args); \
        }                                                                                                          \
        if (LPCT == 16 && sb == 2)                                                                                 \
            return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, 2, false>, dm, lds, stream, "alo_msda_forward", args); \
        if (LPCT == 16 && sb == 8)                                                                                 \
            return launch(msda_fwd_kernel<T, LT, CT, VEC, G, LPCT, (LPCT ? 8 : 2), false>, dm, lds, stream, "alo_msda_forward", args); \
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
    CASE(T, LT, CT, VECW, 64, 0) CASE(T, LT, CT, 1, 8, 0) CASE(T, LT, CT, 1, 64, 0)

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
template <typename K>
int launch_tile(K kernel, const TileArgs& ta, size_t lds, hipStream_t stream) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_forward(tile): %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(ta.nblocks), dim3(kThreads), lds, stream, ta);
    return check_launch("alo_msda_forward(tile)");
}

int forward_impl(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index, const void* loc,
                 const void* attn, const void* ref, int ref_dim, const int32_t* shapes_host, void* out, int N, int S,
                 int M, int D, int L, int Lq, int P, int value_dtype, int loc_dtype, void* stream_) {
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
    const int sb = tuning().fwd_batch;
    if (tuning().tile && value_dtype != ALO_F64) {
        TileArgs ta;
        if (tile_plan(ta, shapes_host, N, S, M, D, L, Lq, P, elem, aligned)) {
            ta.value = value; ta.loc = loc; ta.attn = attn; ta.ref = ref; ta.out = out;
            ta.N = N; ta.S = S; ta.M = M; ta.D = D; ta.Lq = Lq; ta.ref_dim = ref_dim;
            ta.window_budget = tuning().tile_lds;
            const size_t lds = 64 + (size_t)ta.window_budget;  // bbox + [zero row | staged windows]
            if (value_dtype == ALO_BF16)
                return fused ? launch_tile(msda_fwd_tile_kernel<bf16_t, float, float, 8, 4, true>, ta, lds, stream)
                             : launch_tile(msda_fwd_tile_kernel<bf16_t, float, float, 8, 4, false>, ta, lds, stream);
            return fused ? launch_tile(msda_fwd_tile_kernel<float, float, float, 4, 8, true>, ta, lds, stream)
                         : launch_tile(msda_fwd_tile_kernel<float, float, float, 4, 8, false>, ta, lds, stream);
        }
    }
    void* args[] = {&value, &spatial_shapes, &level_start_index, &loc, &attn, &ref, &out, &dm};
    if (value_dtype == ALO_F32) { ALO_ALL_CASES(ALO_FWD_CASE, float, float, float, 4) }
    if (value_dtype == ALO_F64) { ALO_ALL_CASES(ALO_FWD_CASE, double, double, double, 2) }
    if (value_dtype == ALO_BF16) { ALO_ALL_CASES(ALO_FWD_CASE, bf16_t, float, float, 8) }
    return fail(ALO_ERR_UNSUPPORTED, "alo_msda_forward: no kernel for vec=%d group=%d", plan.vec, plan.g);
}
}  // namespace

extern "C" int alo_msda_forward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                const void* sampling_loc, const void* attn_weight, void* out, int N, int S, int M,
                                int D, int L, int Lq, int P, int value_dtype, int loc_dtype, void* stream_) {
    return forward_impl(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, nullptr, 0, nullptr, out, N,
                        S, M, D, L, Lq, P, value_dtype, loc_dtype, stream_);
}

extern "C" int alo_msda_forward_fused(const void* value, const int32_t* spatial_shapes,
                                      const int32_t* level_start_index, const void* sampling_offsets,
                                      const void* attn_logits, const void* reference_points,
                                      const int32_t* spatial_shapes_host, void* out, int N, int S, int M, int D, int L,
                                      int Lq, int P, int ref_dim, int value_dtype, void* stream_) {
    ALO_REQUIRE(reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_msda_forward_fused: reference_points is null");
    ALO_REQUIRE(ref_dim == 2 || ref_dim == 4, ALO_ERR_INVALID_ARGUMENT,
                "alo_msda_forward_fused: last dim of reference_points must be 2 or 4, got %d", ref_dim);
    // the geometry dtype is implied: fp64 for fp64 values, fp32 otherwise (loc_dtype only steers validation here)
    const int loc_dtype = value_dtype == ALO_F64 ? ALO_F64 : ALO_F32;
    return forward_impl(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                        ref_dim, spatial_shapes_host, out, N, S, M, D, L, Lq, P, value_dtype, loc_dtype, stream_);
}

extern "C" int alo_msda_backward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                 const void* sampling_loc, const void* attn_weight, const void* grad_out,
                                 void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int N, int S, int M,
                                 int D, int L, int Lq, int P, int value_dtype, int loc_dtype, void* stream_) {
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
    const Plan plan = make_plan(D, L, P, elem, aligned);
    Dims dm = make_dims(N, S, M, D, L, Lq, P, plan.g);
    void* args[] = {&value, &spatial_shapes, &level_start_index, &sampling_loc, &attn_weight, &grad_out,
                    &grad_value, &grad_sampling_loc, &grad_attn_weight, &dm};
    if (value_dtype == ALO_F32) { ALO_ALL_CASES(ALO_BWD_CASE, float, float, float, 4) }
    if (value_dtype == ALO_F64) { ALO_ALL_CASES(ALO_BWD_CASE, double, double, double, 2) }
    if (value_dtype == ALO_BF16) { ALO_ALL_CASES(ALO_BWD_CASE, bf16_t, float, float, 8) }
    return fail(ALO_ERR_UNSUPPORTED, "alo_msda_backward: no kernel for vec=%d group=%d", plan.vec, plan.g);
}
