// Multi-scale deformable attention, backward — the WIDE path: one workgroup owns a 16x16 block of encoder queries of one head.
//
// Replaces (with msda.hip's kernels) alonet_custom::ms_deform_attn_backward: /root/reference
// alonet/deformable_detr/ops/src/cuda/ms_deform_attn_cuda.cu:83-153, cuda/ms_deform_im2col_cuda.cuh:87-159 (bilinear
// adjoint), :301-403 (one thread per (query, head, level, point), one atomicAdd per corner and channel).
//
// Why it exists.  The chip retires ~10.5 G 128-byte atomic rows per second whatever the pattern (tools/micro/atomic_scope.hip), so
// the time of a scatter-formulated backward is its number of atomic rows / 10.5 G.  msda_bwd_tiled_kernel forms the sums of a
// 4x4 query tile on chip (one row per touched pixel per tile): 6.3 M rows on the random-init ring, but 12-13 M on the survey and
// trained-like spreads, where neighbouring queries share fewer pixels per tile (profiles/r05_bwd_rows_sim.txt).  A 16x16 block
// shares 3-4 x more (2.9 / 4.0 / 4.3 M rows).  The sums of a block this wide do not fit a dense A[row][query] matrix, and LDS float
// atomics are slow (ds_add_f32: 194 clocks per wave instruction, tools/micro/lds_atomic.hip) — so the block SORTS its corners by
// destination pixel instead (a counting sort in LDS):
//
//   per target level (4 passes over the block's 256 queries x 4 points x 4 corners = 4096 corner entries), 512 threads:
//   1. count    thread (query, point pair) turns its two sampling points into taps.  A corner inside the block's WINDOW on that level
//               (the block's footprint + 13 px either side, at most 55 x 55 pixels, rows numbered 64 y + x) takes a slot in its window
//               row: ds_add_rtn_u32 on cnt[row] (integer LDS atomics are fast), and the scalar tbl[entry] = bilinear weight x
//               attention weight.
//   2. scan     exclusive prefix sum of the counts in place (DPP wave scan + 8 wave totals), and the gather's WORK ITEMS: one per
//               touched row, up to four for rows longer than 128 entries — no half wave is handed a very long row alone (splitting at 32 cost
//               40 % more atomic rows on the ring: the items are dealt round-robin, which balances the coarse levels by itself).
//   3. list     list[cnt[row] + slot] = entry: the entries sorted by window row, contiguous per row.
//   4. gather   a QUARTER wave per work item at D = 32 (a half wave at D = 64): four lane groups x K lanes x 8 channels, four entries a
//               step.  The lanes hold value[row] (read from memory once per block, the next item's row prefetched) and per entry ONE
//               D-channel LDS read of the query's grad_out row feeds both sums:
//                 grad_value[row] += tbl[entry] * grad_out[q]      registers; reduce-scatter over the 4 lane groups (ds_swizzle / DPP),
//                                                                  then two buffer_atomic_add_f32 per row, each a contiguous half row
//                                                                  (costs what one whole-row instruction costs: tools/micro/atomic_split.hip)
//                 d[entry]         = <value[row], grad_out[q]>      2-3 DPP adds over the K lanes; overwrites tbl[entry]
//   5. finish   the thread that owns the sample reads its four d's back: grad_attn = sum_k w_k d_k, grad_loc = attn * (W, H) * (...)
//               — the same expressions as msda_bwd_tiled_kernel's stage 3 — and stores 16 + 8 contiguous bytes.
//   A sample with a corner outside the window (far outliers; every sample of a uniform distribution) sets a bit in an overflow mask and
//   takes the per-corner route of msda_bwd_kernel at the end: same results, old cost.
//
// Measured (MI355X, N = 4 encoder call, fp32; tools/exp/bwd_wide_check.py): 0.58 / 0.66 / 0.66 / 3.15 ms on the ring / survey /
// trained-like / uniform distributions against 0.70 / 1.32 / 1.61 / 4.04 for msda_bwd_tiled_kernel; bf16 values 0.59 / 0.67 against
// 4.1 / 3.9 for msda_bwd_kernel; D = 64 1.36 / 1.55 against 8.4 / 7.8.  VALU-issue bound (67 % of the SIMD time): the build log with every
// intermediate number is docs/experiments.md R6.1.
//
// Shapes served: value / grad_out fp32 or bf16 (gradients fp32), D = 32, L = P = 4, queries = the pyramid's own pixels (Lq == S).
// The host copy of the shapes sizes the grid AND travels by value; a launch whose device shapes differ from it sends every
// sample down the per-corner route with the queries grouped 256 in a row (correct, slow — the Python host never caches shapes).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace alo {
namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Parts of the kernel can be switched off for timing in a development build (results are then wrong on purpose): 1 no flush, 2 no
// list walk, 4 no value rows, 8 no insert, 16 no gradient stores, 32 no grad_out rows, 64 no locations, 128 no gather, 1024 no scan.
#ifdef ALO_WIDE_DBG
#define ALO_DBG(bit) ((wd.dbg & (bit)) != 0)
#else
#define ALO_DBG(bit) (false)
#endif

constexpr int kSlots = 256;              // query slots of a block (16 x 16 on the fine levels)
constexpr int kWThreads = 512;           // 8 waves: thread = (query slot, point pair)
constexpr int kSeg = 128;                 // entries of a row one work item walks: 32 / 64 / 128 / 512 measured 0.68 / 0.640 / 0.640 / 0.639 ms (ring)
constexpr int kClip = 56;                // window side limit (55 in use): footprint (<= 32) + ~12 px of halo either side
constexpr unsigned kDrop = 0xffffff00u;  // a byte offset past any frame slab: the buffer range check drops the lane

constexpr int kCnt = 55 * 64 + 8;                     // window rows are numbered 64 * y + x (y, x < 55): no division on the way back
constexpr int kOffTbl = 0;                            // float  tbl[4096]       w * attn per corner entry, then d
constexpr int kOffCnt = kOffTbl + 4096 * 4;           // u32    cnt[kCnt]       entries per window row, then their exclusive prefix sum
constexpr int kOffList = kOffCnt + kCnt * 4;          // u16    list[4096 + 32] entries sorted by window row
constexpr int kOffOvf = kOffList + (4096 + 32) * 2;   // u32    ovf[128]        bit per (level, sample): takes the per-corner route
constexpr int kOffItems = kOffOvf + 128 * 4;          // u16    items[3168]     work items of the gather: row | (segment << 12)
constexpr int kOffMisc = kOffItems + 3168 * 2;        // u32    wsum[8], total  wave totals of the prefix sum
constexpr int kOffG = kOffMisc + 64;                  // float  G[256][D]       grad_out rows of the block's queries
constexpr int wide_lds(int D) { return kOffG + kSlots * D * 4; }   // 78,416 bytes at D = 32 (two workgroups per CU), 111,184 at D = 64 (one)

struct WideDims {
    int N, S, M, Lq;
    int h[4], w[4];     // host copy of the shapes
    int start[4];       // first pixel of every level, from the host copy
    int sh[4];          // log2 of the block side per level (footprint on the finest level <= 32 px)
    int nbx[4];         // blocks per row of blocks
    int first[5];       // first block of every level; first[4] = blocks per (batch item, head)
    unsigned nblocks;
    int dbg;            // timing experiments (a build with -DALO_WIDE_DBG reads the environment variable ALO_WIDE_DBG; tools/exp/bwd_wide_dbg.py)
};

template <typename T>
struct Row4;   // four consecutive channels of a value / grad_out row -> fp32
template <>
struct Row4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t r, unsigned elem_off) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, elem_off == kDrop ? kDrop : elem_off * 4u, 0, 0);
        return f32x4{__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)};
    }
    static __device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t r, unsigned elem_off) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, elem_off == kDrop ? kDrop : elem_off * 4u, 0, 0));
    }
    // eight channels at a BYTE offset (kDrop, and kDrop + 16, are past the slab: zeros)
    template <int CH1>   // the second four channels start CH1 channels after the first
    static __device__ __forceinline__ void load8b(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4& lo, f32x4& hi) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0), y = __builtin_amdgcn_raw_buffer_load_b128(r, off + 4u * CH1, 0, 0);
        lo = f32x4{__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)};
        hi = f32x4{__uint_as_float(y.x), __uint_as_float(y.y), __uint_as_float(y.z), __uint_as_float(y.w)};
    }
};
template <>
struct Row4<bf16_t> {
    static __device__ __forceinline__ f32x4 widen(u32x2 x) {
        return f32x4{__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xffff0000u), __uint_as_float(x.y << 16),
                     __uint_as_float(x.y & 0xffff0000u)};
    }
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) { return widen(*reinterpret_cast<const u32x2*>(p)); }
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t r, unsigned elem_off) {
        return widen(__builtin_amdgcn_raw_buffer_load_b64(r, elem_off == kDrop ? kDrop : elem_off * 2u, 0, 0));
    }
    static __device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t r, unsigned elem_off) {
        return bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r, elem_off == kDrop ? kDrop : elem_off * 2u, 0, 0));
    }
    template <int CH1>
    static __device__ __forceinline__ void load8b(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4& lo, f32x4& hi) {
        if constexpr (CH1 == 4) {
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            lo = widen(u32x2{x.x, x.y});
            hi = widen(u32x2{x.z, x.w});
        } else {
            lo = widen(__builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
            hi = widen(__builtin_amdgcn_raw_buffer_load_b64(r, off + 2u * CH1, 0, 0));
        }
    }
};

template <int CTRL>
__device__ __forceinline__ float dppc(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// the value of lane ^ X (X = 4, 8, 16)
template <int X>
__device__ __forceinline__ float lane_xor(float v) {
    if constexpr (X == 8) {
        return dppc<0x128>(v);   // row_ror:8 inside a 16-lane row
    } else {
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), X == 4 ? 0x101F : 0x401F));   // xor_mask << 10 | and_mask 0x1f
    }
}
// sum over aligned groups of 8 lanes, valid in every lane of the group
__device__ __forceinline__ float sum8(float v) {
    v += dppc<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dppc<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dppc<0x141>(v);   // row_half_mirror: lane i <-> 7 - i inside each half row
    return v;
}
// sum over each 32-lane half; valid in lanes 16-31 / 48-63
__device__ __forceinline__ float half_sum(float v) {
    v += dppc<0x128>(v);
    v += dppc<0x124>(v);
    v += dppc<0x122>(v);
    v += dppc<0x121>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    return v;
}

struct Tap {
    int h_low, w_low;
    float lh, lw;
    unsigned flags;   // bit k: corner k inside the map; bit 4: the sample counts (cuh:285-291, :38-78)
};
__device__ __forceinline__ Tap make_tap_w(float x, float y, int Hl, int Wl, bool live) {
    Tap t;
    const float h_im = y * (float)Hl - 0.5f, w_im = x * (float)Wl - 0.5f;
    const bool valid = live && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
    const float hs = valid ? h_im : 0.f, ws = valid ? w_im : 0.f;
    const float hf = floorf(hs), wf = floorf(ws);
    t.h_low = (int)hf;
    t.w_low = (int)wf;
    t.lh = hs - hf;
    t.lw = ws - wf;
    const bool hl = t.h_low >= 0, hh = t.h_low + 1 <= Hl - 1, wl = t.w_low >= 0, wh = t.w_low + 1 <= Wl - 1;
    t.flags = valid ? ((hl && wl ? 1u : 0u) | (hl && wh ? 2u : 0u) | (hh && wl ? 4u : 0u) | (hh && wh ? 8u : 0u) | 16u) : 0u;
    return t;
}

#define ALO_WAVE_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

template <typename T, int D>
__global__ void __launch_bounds__(kWThreads, D == 32 ? 4 : 2)
msda_bwd_wide_kernel(const T* __restrict__ value, const int32_t* __restrict__ shapes, const int32_t* __restrict__ lstart,
                     const float* __restrict__ loc, const float* __restrict__ attn, const T* __restrict__ grad_out,
                     float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
                     const WideDims wd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* G = reinterpret_cast<float*>(smem + kOffG);
    float* tbl = reinterpret_cast<float*>(smem + kOffTbl);
    unsigned* cnt = reinterpret_cast<unsigned*>(smem + kOffCnt);
    unsigned short* list = reinterpret_cast<unsigned short*>(smem + kOffList);
    unsigned* ovf = reinterpret_cast<unsigned*>(smem + kOffOvf);
    unsigned short* items = reinterpret_cast<unsigned short*>(smem + kOffItems);
    unsigned* wsum = reinterpret_cast<unsigned*>(smem + kOffMisc);

    if (ALO_DBG(256)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = wd.M, S = wd.S, Lq = wd.Lq;
    // launch order: head fastest (block i runs on XCD i % 8: with M = 8 every XCD serves one head), then the batch item, then the
    // block — the full 16x16 blocks of the finest level of EVERY item first, the small blocks of the coarse levels fill the tail
    const unsigned lb = blockIdx.x;
    const int m = lb % M;
    const int b = (int)((lb / M) % (unsigned)wd.N);
    const int blk = (int)(lb / ((unsigned)M * wd.N));

    // the device copy of the geometry must be the host's: the block table below was sized from the host's
    bool same = true;
#pragma unroll
    for (int l = 0; l < 4; ++l)
        same = same && shapes[2 * l] == wd.h[l] && shapes[2 * l + 1] == wd.w[l] && lstart[l] == wd.start[l];
    if (!same && blk * kSlots >= Lq) return;   // linear grouping: ceil(Lq / 256) blocks hold every query (block-uniform exit)

    int ls = 0;
#pragma unroll
    for (int l = 1; l < 4; ++l)
        if (blk >= wd.first[l]) ls = l;
    const int shq = wd.sh[ls];
    const int by = (blk - wd.first[ls]) / wd.nbx[ls], bx = (blk - wd.first[ls]) - by * wd.nbx[ls];
    const int Hq = wd.h[ls], Wq = wd.w[ls], Sq = wd.start[ls];
    auto query_of = [&](int slot) -> int {   // slot of the block -> query index, -1 when the slot is empty
        if (same) {
            const int qy = (by << shq) + (slot >> shq), qx = (bx << shq) + (slot & ((1 << shq) - 1));
            return (slot < (1 << (2 * shq)) && qy < Hq && qx < Wq) ? Sq + qy * Wq + qx : -1;
        }
        const int q = blk * kSlots + slot;
        return q < Lq ? q : -1;
    };
    const long bq0 = (long)b * Lq;

    // ---- set-up: grad_out rows of the block, empty counters ----------------------------------------------------------------
    constexpr int CH = D / 4;   // 16-byte chunks of a grad_out row
#pragma unroll
    for (int j = 0; j < kSlots * CH / kWThreads; ++j) {
        const int slot = tid / CH + (kWThreads / CH) * j, c4 = tid % CH;
        const int q = query_of(slot);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (q >= 0 && !ALO_DBG(32)) g = Row4<T>::load(grad_out + ((bq0 + q) * M + m) * D + 4 * c4);
        *reinterpret_cast<f32x4*>(G + slot * D + 4 * c4) = g;
    }
    for (int r = tid; r < kCnt; r += kWThreads) cnt[r] = 0;
    if (tid < 128) ovf[tid] = same ? 0u : 0xffffffffu;
    // the walk reads up to 15 entries past an item's end: every list slot always holds a valid entry number (weights are masked, the
    // grad_out row it names is finite)
    for (int i = tid; i < (4096 + 32) / 2; i += kWThreads) reinterpret_cast<unsigned*>(list)[i] = 0;

    const int slot = tid >> 1, ph = tid & 1;   // neighbouring lanes hold the two point pairs of a query: 32 contiguous bytes of loc / grad_loc
    const int q_own = query_of(slot);
    const bool live = q_own >= 0;
    const long qm = (bq0 + (live ? q_own : 0)) * M + m;

    const size_t slab = (size_t)b * S * M * D;
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(value + slab, (unsigned)((size_t)S * M * D * sizeof(T)));
    const __amdgpu_buffer_rsrc_t gv_rsrc = make_rsrc(grad_value + slab, (unsigned)((size_t)S * M * D * 4));
    const unsigned head_elems = (unsigned)m * D, pix_elems = (unsigned)M * D;

    // normalised centre of the block on its own level: the window of every level is a clip box around it
    const float cxn = ((float)(bx << shq) + 0.5f * (float)(1 << shq)) / (float)Wq;
    const float cyn = ((float)(by << shq) + 0.5f * (float)(1 << shq)) / (float)Hq;

    f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
    float a2[2] = {0.f, 0.f};
    if (live && same && !ALO_DBG(64)) {
        l4 = *reinterpret_cast<const f32x4*>(loc + (qm * 4 + 0) * 8 + ph * 4);
        const f32x2_t av = *reinterpret_cast<const f32x2_t*>(attn + (qm * 4 + 0) * 4 + ph * 2);
        a2[0] = av[0];
        a2[1] = av[1];
    }
    __syncthreads();

    if (ALO_DBG(512)) return;
    if (same) {
        for (int lt = 0; lt < 4; ++lt) {
            const int Hl = wd.h[lt], Wl = wd.w[lt], Sl = wd.start[lt];
            // window: the block's footprint on this level + 13 px either side, at most 55 x 55
            const int cxl = (int)floorf(cxn * (float)Wl), cyl = (int)floorf(cyn * (float)Hl);
            const int fx = ((Wl << shq) + Wq - 1) / Wq, fy = ((Hl << shq) + Hq - 1) / Hq;
            const int hx = min((fx >> 1) + 13, kClip / 2 - 1), hy = min((fy >> 1) + 13, kClip / 2 - 1);
            const int wx0 = max(cxl - hx, 0), wx1 = min(cxl + hx, Wl - 1);
            const int wy0 = max(cyl - hy, 0), wy1 = min(cyl + hy, Hl - 1);
            const int wh = wy1 - wy0 + 1;
            const int rows = wh * 64;   // row numbers in use: 64 * y + x, x < ww <= 55

            // ---- 1. count: every in-window corner takes a slot in its row ---------------------------------------------------------
            Tap t[2];
            bool inwin[2];
            unsigned rs[2][4];   // (row << 16) | slot of the corner in its row
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t[j] = make_tap_w(l4[2 * j], l4[2 * j + 1], Hl, Wl, live);
                if (ALO_DBG(8)) t[j].flags = 0;
                // every in-map corner inside the window?  (in-map corners have coordinates in [0, W-1] x [0, H-1])
                const int xa = max(t[j].w_low, 0), xb = min(t[j].w_low + 1, Wl - 1);
                const int ya = max(t[j].h_low, 0), yb = min(t[j].h_low + 1, Hl - 1);
                inwin[j] = xa >= wx0 && xb <= wx1 && ya >= wy0 && yb <= wy1;
                const int s = slot * 4 + ph * 2 + j;
                if (t[j].flags & 16u) {
                    if (inwin[j]) {
                        const float hh = 1.f - t[j].lh, hw = 1.f - t[j].lw, at = a2[j];
                        const float w4[4] = {hh * hw * at, hh * t[j].lw * at, t[j].lh * hw * at, t[j].lh * t[j].lw * at};
                        const int base = (t[j].h_low - wy0) * 64 + (t[j].w_low - wx0);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if ((t[j].flags >> k) & 1u) {
                                const int row = base + (k >> 1) * 64 + (k & 1);
                                const unsigned pos = __hip_atomic_fetch_add(&cnt[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                rs[j][k] = ((unsigned)row << 16) | pos;
                            }
                        *reinterpret_cast<f32x4*>(tbl + s * 4) = f32x4{w4[0], w4[1], w4[2], w4[3]};   // entries of out-of-map corners are never read
                    } else {
                        __hip_atomic_fetch_or(&ovf[lt * 32 + (s >> 5)], 1u << (s & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            // the next level's locations and weights travel while this level is sorted and gathered
            f32x4 l4n = {0.f, 0.f, 0.f, 0.f};
            float a2n[2] = {0.f, 0.f};
            if (live && lt < 3 && !ALO_DBG(64)) {
                l4n = *reinterpret_cast<const f32x4*>(loc + (qm * 4 + lt + 1) * 8 + ph * 4);
                const f32x2_t av = *reinterpret_cast<const f32x2_t*>(attn + (qm * 4 + lt + 1) * 4 + ph * 2);
                a2n[0] = av[0];
                a2n[1] = av[1];
            }
            __syncthreads();

            // ---- 2. exclusive prefix sum of the counts, in place (cnt[rows] = number of entries), and the gather's work items: one per
            //         kSeg entries of a row, so that no half wave is handed a very long row alone -------------------------------------------------
            if (!ALO_DBG(1024)) {
                const int per = (rows + kWThreads) / kWThreads;   // rows + 1 counters over 512 threads: at most 7 each
                const int r0 = tid * per;
                unsigned c7[7], sum = 0;   // low half: entries; high half: work items
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int r = r0 + i;
                    const unsigned n = (i < per && r < rows) ? cnt[r] : 0u;
                    c7[i] = n | (min((n + (unsigned)kSeg - 1u) / (unsigned)kSeg, 4u) << 16);
                    sum += c7[i];
                }
                unsigned incl = sum;
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
                incl += (unsigned)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                unsigned base = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) base += w < wave ? wsum[w] : 0u;
                unsigned excl = base + incl - sum;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int r = r0 + i;
                    if (i < per && r <= rows) cnt[r] = excl & 0xffffu;
                    const unsigned ni = c7[i] >> 16, ib = excl >> 16;
                    if (ni) items[ib] = (unsigned short)r;
                    if (ni > 1) {   // long rows (coarse levels): up to three more segments, the last one takes whatever is left
                        items[ib + 1] = (unsigned short)(r | 0x1000);
                        if (ni > 2) items[ib + 2] = (unsigned short)(r | 0x2000);
                        if (ni > 3) items[ib + 3] = (unsigned short)(r | 0x3000);
                    }
                    excl += c7[i];
                }
                if (tid == kWThreads - 1) wsum[8] = excl >> 16;   // number of work items
            }
            __syncthreads();

            // ---- 3. the sorted list ---------------------------------------------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if ((t[j].flags & 16u) && inwin[j]) {
                    const int s = slot * 4 + ph * 2 + j;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((t[j].flags >> k) & 1u) list[cnt[rs[j][k] >> 16] + (rs[j][k] & 0xffffu)] = (unsigned short)(s * 4 + k);
                }
            __syncthreads();

            // ---- 4. gather: LPI lanes per work item (a row, or kSeg entries of a very long row): a quarter wave at D = 32, a half wave at
            //         D = 64; K lanes x 8 channels per entry, 4 entries of an item a step -------------------------------------------------
            {
                // Lane c of an entry's K lanes holds channels 4c .. 4c+3 of EACH half of the row: after the reduce-scatter over the item's
                // four lane groups every lane owns one channel of each half, and each of the two flush instructions covers a contiguous
                // 64 bytes (D = 32) / 128 bytes (D = 64) per row — which costs what a whole-row instruction costs (tools/micro/atomic_split.hip:
                // 10.4 G rows/s either way; the same dwords spread over the whole row: half of it)
                constexpr int K = D / 8, NG = 4, LPI = K * NG, IPW = 64 / LPI, CH1 = D / 2;
                const int sub = lane / LPI, g = (lane / K) & (NG - 1), c = lane & (K - 1);
                const int n_items = ALO_DBG(128) ? 0 : (int)wsum[8];
                const unsigned ch0 = 4u * c;
                const unsigned lane_b = (head_elems + ch0) * (unsigned)sizeof(T);   // byte offset of the lane's first 4 channels inside a pixel
                const unsigned pix_b = pix_elems * (unsigned)sizeof(T);
                const unsigned base_pix = (unsigned)(Sl + wy0 * Wl + wx0);
                // item k -> byte offset of (row's pixel, this lane's channels) in `value`; first entry and entry count in `fl`
                auto item_of = [&](int k, unsigned& fl) -> unsigned {
                    fl = 0;
                    if (k >= n_items) return kDrop;
                    const unsigned it = items[k];
                    const unsigned row = it & 0xfffu, sg = it >> 12;
                    const unsigned o0 = cnt[row] + sg * (unsigned)kSeg, o1 = cnt[row + 1];
                    const unsigned len = sg == 3u ? o1 - o0 : min(o1 - o0, (unsigned)kSeg);
                    fl = o0 | (len << 16);
                    const unsigned pix = base_pix + __umul24(row >> 6, (unsigned)Wl) + (row & 63u);
                    return __umul24(pix, pix_b) + lane_b;
                };
                unsigned fl;
                unsigned voff = item_of(wave * IPW + sub, fl);
                f32x4 v0, v1;
                Row4<T>::template load8b<CH1>(v_rsrc, ALO_DBG(4) ? kDrop : voff, v0, v1);
                const float* Gc = G + ch0;
                const bool t1 = g & 1, t2 = g & 2;
                const unsigned reg_b = ((t1 ? 2u : 0u) + (t2 ? 1u : 0u)) * 4u;   // the channel (of its four per half) the lane flushes
                for (int kb = wave * IPW; kb < n_items; kb += 8 * IPW) {   // wave-uniform bound: the wave's items are kb .. kb + IPW - 1
                    // the next item's value row travels while this one is walked
                    unsigned fl_n;
                    const unsigned voff_n = item_of(kb + sub + 8 * IPW, fl_n);
                    f32x4 vn0, vn1;
                    Row4<T>::template load8b<CH1>(v_rsrc, ALO_DBG(4) ? kDrop : voff_n, vn0, vn1);
                    const int first = (int)(fl & 0xffffu), len = ALO_DBG(2) ? 0 : (int)(fl >> 16);
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                    int longest = max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 32));
                    if constexpr (IPW == 4) longest = max(longest, max(__builtin_amdgcn_readlane(len, 16), __builtin_amdgcn_readlane(len, 48)));
                    const int steps = (longest + NG - 1) / NG;   // wave-uniform
                    // (a slot past the item's end re-reads the row's OWN first entry with weight 0: a non-finite grad_out row of some other
                    // query can then never reach this row — 0 x inf would — exactly as in the reference, where it touches only its own pixels)
                    auto walk = [&](int s0, auto nsteps) {
                        constexpr int NS = decltype(nsteps)::value;
                        unsigned e[NS];
                        bool ok[NS];
                        float w[NS];
                        f32x4 g0[NS], g1[NS];
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            const int i = (s0 + u) * NG + g;
                            ok[u] = i < len;
                            e[u] = list[first + (ok[u] ? i : 0)];   // idle slots re-read the item's first entry (weight forced to 0)
                        }
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            w[u] = tbl[e[u]];
                            const float* gp = Gc + (e[u] >> 4) * D;
                            g0[u] = *reinterpret_cast<const f32x4*>(gp);
                            g1[u] = *reinterpret_cast<const f32x4*>(gp + CH1);
                        }
                        float dp[NS];
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            const float wu = ok[u] ? w[u] : 0.f;
                            a0 += wu * g0[u];
                            a1 += wu * g1[u];
                            const f32x4 pr = v0 * g0[u] + v1 * g1[u];
                            float d = (pr[0] + pr[1]) + (pr[2] + pr[3]);
                            d += dppc<0xB1>(d);    // quad_perm [1,0,3,2]
                            d += dppc<0x4E>(d);    // quad_perm [2,3,0,1]
                            if constexpr (K == 8) d += dppc<0x141>(d);   // row_half_mirror: the other quad of the 8 lanes
                            dp[u] = d;
                        }
                        if (c == 0) {
#pragma unroll
                            for (int u = 0; u < NS; ++u)
                                if (ok[u]) tbl[e[u]] = dp[u];
                        }
                    };
                    if (steps == 1) {   // most rows of the fine levels: at most 8 entries
                        walk(0, std::integral_constant<int, 1>{});
                    } else {
                        for (int s0 = 0; s0 < steps; s0 += 2) walk(s0, std::integral_constant<int, 2>{});
                    }
                    // reduce-scatter over the lane groups: every lane ends with the full sum of ONE of its channels (two at D = 64), the 32
                    // lanes of the half wave cover 32 consecutive channels: ONE atomic row per work item and 128 bytes of the row.
                    // Lane ^ 4 and lane ^ 16 travel on the LDS crossbar (ds_swizzle: no memory, no address register), lane ^ 8 on DPP.
                    // value is T, grad_value fp32: the same pixel and channels are at byte offset (voff / sizeof(T)) * 4
                    const unsigned boff = (voff == kDrop || ALO_DBG(1)) ? kDrop : voff * (4u / (unsigned)sizeof(T)) + reg_b;
                    {
                        // four groups: first the pair of channels, then the channel — of BOTH halves of the row at once.  The partner groups
                        // are K and 2 K lanes away: lane ^ 4 / lane ^ 16 travel on the LDS crossbar (ds_swizzle: no memory, no address
                        // register), lane ^ 8 on DPP (row_ror:8 inside a 16-lane row)
                        auto xchg1 = [](float x) { return lane_xor<K>(x); };
                        auto xchg2 = [](float x) { return lane_xor<2 * K>(x); };
                        const float ka = t1 ? a0[2] : a0[0], kb2 = t1 ? a0[3] : a0[1], kc = t1 ? a1[2] : a1[0], kd = t1 ? a1[3] : a1[1];
                        const float sa = t1 ? a0[0] : a0[2], sb = t1 ? a0[1] : a0[3], sc = t1 ? a1[0] : a1[2], sd = t1 ? a1[1] : a1[3];
                        const float pa = ka + xchg1(sa), pb = kb2 + xchg1(sb), pc = kc + xchg1(sc), pd = kd + xchg1(sd);
                        const float k0 = t2 ? pb : pa, k1 = t2 ? pd : pc, x0 = t2 ? pa : pb, x1 = t2 ? pc : pd;
                        const float tot0 = k0 + xchg2(x0), tot1 = k1 + xchg2(x1);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tot0, gv_rsrc, boff, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tot1, gv_rsrc, boff == kDrop ? kDrop : boff + 4u * CH1, 0, 0);
                    }
                    voff = voff_n;
                    fl = fl_n;
                    v0 = vn0;
                    v1 = vn1;
                }
            }
            __syncthreads();

            // ---- 5. finish: the sample's owner combines its four d's; the counters are cleared for the next level -------------------------
            if (live && !ALO_DBG(16)) {
                f32x4 gl = {0.f, 0.f, 0.f, 0.f};
                f32x2_t ga = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int s = slot * 4 + ph * 2 + j;
                    if ((t[j].flags & 16u) && inwin[j]) {   // (samples on the per-corner route are written again at the end, after this store)
                        const f32x4 d4 = *reinterpret_cast<const f32x4*>(tbl + s * 4);
                        float dk[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) dk[k] = (t[j].flags >> k) & 1u ? d4[k] : 0.f;
                        const float lh = t[j].lh, lw = t[j].lw, hh = 1.f - lh, hw = 1.f - lw;
                        ga[j] = hh * hw * dk[0] + hh * lw * dk[1] + lh * hw * dk[2] + lh * lw * dk[3];
                        const float gww = -hh * dk[0] + hh * dk[1] - lh * dk[2] + lh * dk[3];
                        const float ghw = -hw * dk[0] - lw * dk[1] + hw * dk[2] + lw * dk[3];
                        gl[2 * j] = (float)Wl * gww * a2[j];
                        gl[2 * j + 1] = (float)Hl * ghw * a2[j];
                    }
                }
                const long gidx = (qm * 4 + lt) * 4 + ph * 2;
                *reinterpret_cast<f32x2_t*>(grad_attn + gidx) = ga;
                *reinterpret_cast<f32x4*>(grad_loc + 2 * gidx) = gl;
            }
            if (!ALO_DBG(2048)) for (int r = tid; r <= rows; r += kWThreads) cnt[r] = 0;
            l4 = l4n;
            a2[0] = a2n[0];
            a2[1] = a2n[1];
            __syncthreads();
        }
    }

    // ---- 6. the per-corner route: one sample per half wave, one 128-byte row per corner (msda_bwd_kernel's way) ---------------------
    {
        const int ch = lane & 31, half = lane >> 5;
        for (int wi = wave; wi < 128; wi += 8) {
            unsigned word = ovf[wi];   // wave-uniform
            word = __builtin_amdgcn_readfirstlane(word);
            while (word) {
                const int b0 = __builtin_ctz(word);
                word &= word - 1;
                int b1 = -1;
                if (word) { b1 = __builtin_ctz(word); word &= word - 1; }
                const int bit = half ? b1 : b0;
                const int lv = wi >> 5, s = ((wi & 31) << 5) + bit, sl = s >> 2, p = s & 3;
                const int q = bit >= 0 ? query_of(sl) : -1;
                const int Hlv = shapes[2 * lv], Wlv = shapes[2 * lv + 1], Slv = lstart[lv];
                float s_attn = 0.f, s_w = 0.f, s_h = 0.f;
                long gidx = 0;
                if (q >= 0) {
                    gidx = (((bq0 + q) * M + m) * 4 + lv) * 4 + p;
                    const float x = loc[2 * gidx], y = loc[2 * gidx + 1], at = attn[gidx];
                    const Tap tp = make_tap_w(x, y, Hlv, Wlv, true);
                    if (tp.flags & 16u) {
                        const float lh = tp.lh, lw = tp.lw, hh = 1.f - lh, hw = 1.f - lw;
                        const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                        const long pix0 = (long)Slv + (long)tp.h_low * Wlv + tp.w_low;
                        const long px[4] = {pix0, pix0 + 1, pix0 + Wlv, pix0 + Wlv + 1};
#pragma unroll
                        for (int c0 = 0; c0 < D; c0 += 32) {   // 32 channels per half wave at a time
                            const float top = G[sl * D + c0 + ch], tgv = top * at;
                            float v[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                v[k] = Row4<T>::load1(v_rsrc, (tp.flags >> k) & 1u ? (unsigned)px[k] * pix_elems + head_elems + c0 + ch : kDrop);
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(
                                    w4[k] * tgv, gv_rsrc, (tp.flags >> k) & 1u ? ((unsigned)px[k] * pix_elems + head_elems + c0 + ch) * 4u : kDrop, 0, 0);
                            s_attn += top * (w4[0] * v[0] + w4[1] * v[1] + w4[2] * v[2] + w4[3] * v[3]);
                            s_w += tgv * (-hh * v[0] + hh * v[1] - lh * v[2] + lh * v[3]);
                            s_h += tgv * (-hw * v[0] - lw * v[1] + hw * v[2] + lw * v[3]);
                        }
                    }
                }
                s_attn = half_sum(s_attn);
                s_w = half_sum(s_w);
                s_h = half_sum(s_h);
                if (ch == 16 && q >= 0) {
                    grad_attn[gidx] = s_attn;
                    grad_loc[2 * gidx] = (float)Wlv * s_w;
                    grad_loc[2 * gidx + 1] = (float)Hlv * s_h;
                }
            }
        }
    }
}

}  // namespace

// Host side of the wide path.  Returns ALO_OK after enqueuing, or ALO_ERR_UNSUPPORTED (nothing enqueued) when the geometry is not
// one the block table can describe — the caller then takes msda_bwd_tiled_kernel.
int msda_backward_wide(const void* value, const int32_t* shapes, const int32_t* lstart, const void* loc, const void* attn,
                       const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn, int N, int S, int M, int D, int Lq,
                       int value_dtype, const int32_t* host_shapes, hipStream_t stream, bool plan_only) {
    if (!host_shapes || Lq != S || (D != 32 && D != 64)) return ALO_ERR_UNSUPPORTED;
    if (value_dtype != ALO_F32 && value_dtype != ALO_BF16) return ALO_ERR_UNSUPPORTED;
    if (S >= (1 << 24) || (long)M * D * 4 >= (1L << 24)) return ALO_ERR_UNSUPPORTED;   // 24-bit multiplies in the gather's address arithmetic
    if ((double)S * M * D * 4 >= 4.0e9) return ALO_ERR_UNSUPPORTED;     // 32-bit byte offsets inside one frame
    WideDims wd;
    wd.N = N; wd.S = S; wd.M = M; wd.Lq = Lq;
    long total = 0;
    int hmax = 1, wmax = 1;
    for (int l = 0; l < 4; ++l) {
        wd.h[l] = host_shapes[2 * l];
        wd.w[l] = host_shapes[2 * l + 1];
        if (wd.h[l] <= 0 || wd.w[l] <= 0 || wd.h[l] >= 32768 || wd.w[l] >= 32768) return ALO_ERR_UNSUPPORTED;
        wd.start[l] = (int)total;
        total += (long)wd.h[l] * wd.w[l];
        hmax = wd.h[l] > hmax ? wd.h[l] : hmax;
        wmax = wd.w[l] > wmax ? wd.w[l] : wmax;
    }
    if (total != S) return ALO_ERR_UNSUPPORTED;
    int blocks = 0;
    for (int l = 0; l < 4; ++l) {
        // the block's footprint on the finest level stays within 32 px (5 % slack: 167 / 84, 100 / 13 are not powers of two)
        const double r = std::max((double)hmax / wd.h[l], (double)wmax / wd.w[l]);
        int sh = 4;
        while (sh > 0 && (double)(1 << sh) * r > 32.0 * 1.05) --sh;
        wd.sh[l] = sh;
        const int bs = 1 << sh;
        wd.nbx[l] = (wd.w[l] + bs - 1) / bs;
        wd.first[l] = blocks;
        blocks += wd.nbx[l] * ((wd.h[l] + bs - 1) / bs);
    }
    wd.first[4] = blocks;
    if (blocks * kSlots < Lq) return ALO_ERR_UNSUPPORTED;   // cannot happen (a block holds at most 256 queries)
    const long nb = (long)N * blocks * M;
    if (nb >= 0x7fffffffL) return ALO_ERR_UNSUPPORTED;
    wd.nblocks = (unsigned)nb;
    if (plan_only) return ALO_OK;
    wd.dbg = getenv("ALO_WIDE_DBG") ? atoi(getenv("ALO_WIDE_DBG")) : 0;
    void* args[] = {&value, &shapes, &lstart, &loc, &attn, &grad_out, &grad_value, &grad_loc, &grad_attn, &wd};
    const int which = (value_dtype == ALO_F32 ? 0 : 1) + (D == 32 ? 0 : 2);
    const void* fns[4] = {reinterpret_cast<const void*>(msda_bwd_wide_kernel<float, 32>), reinterpret_cast<const void*>(msda_bwd_wide_kernel<bf16_t, 32>),
                          reinterpret_cast<const void*>(msda_bwd_wide_kernel<float, 64>), reinterpret_cast<const void*>(msda_bwd_wide_kernel<bf16_t, 64>)};
    static unsigned long long attr_done[4] = {0, 0, 0, 0};   // one bit per device
    hipError_t ea = ensure_dynamic_lds(fns[which], wide_lds(D), &attr_done[which]);
    if (ea != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_backward (wide): %s", hipGetErrorString(ea));
    hipError_t el = hipLaunchKernel(fns[which], dim3(wd.nblocks), dim3(kWThreads), args, wide_lds(D), stream);
    if (el != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_msda_backward (wide): %s", hipGetErrorString(el));
    return check_launch("alo_msda_backward (wide)");
}

}  // namespace alo
