// RAFT all-pairs correlation pyramid + windowed lookup for gfx950 (MI355X).
//
// Reference semantics: alonet/raft/corr.py:13-60 (volume = fmap1^T . fmap2 / sqrt(C); 3x avg_pool2d; 9x9 bilinear
// window per level) and alonet/raft/utils/utils.py:5-19 (pixel -> [-1,1] -> grid_sample(align_corners=True)).
//
// Build: one dense contraction per pair, volume[b,i,j] = <fmap1[b,:,i], fmap2[b,:,j]> / sqrt(C), on the fp16 matrix cores at
//        fp32 accuracy: a pre-pass scales every feature map by a power of two (per batch item, from its largest magnitude, so
//        that nothing overflows or sinks into fp16's subnormals), splits every value into TWO fp16 terms (x = hi + lo, 11 + 11
//        significant bits, round-to-nearest: |x - hi - lo| <= 2^-23 |x|) and lays them out channel-contiguous per pixel; the GEMM
//        accumulates the three cross products hi*hi, hi*lo, lo*hi in fp32 (the dropped lo*lo is <= 2^-22 relative) with
//        v_mfma_f32_32x32x16_f16 - 3 instructions of 32 cycles per 16 channels where the exact-fp32 form
//        (v_mfma_f32_32x32x2_f32) needs 8 of 64, and half of what a three-term bf16 split needs.  The chip runs this kernel at
//        its power limit (1.3 kW measured), so matrix instructions and operand bytes not spent are time not spent.
//        A workgroup's column tile is a 4 x 32 PATCH of the (h2, w2) grid, so the 2x2 and 4x4
//        averages of the pyramid's levels 1 and 2 are sums of accumulators the wave already holds (lane neighbours + its four
//        row tiles): levels 0-2 leave in one launch and the level-0 volume is never re-read.  Levels >= 3 (1.6 % of the
//        columns) are the same contraction against a 2x2-pooled copy of fmap2 (pooling commutes with the inner product).
//
// Lookup: one workgroup serves 32 consecutive query pixels.  Stage 1 reads each (query, level, window-row) strip of
// 2r+3 taps once and interpolates it horizontally into LDS; stage 2 interpolates vertically and writes the
// (B, L*(2r+1)^2, H, W) output with 128-byte contiguous stores per channel.  The reference issues 4 grid_sample launches
// plus meshgrid / cat / permute copies per iteration.
#include "common.hpp"

namespace alo {
namespace {

constexpr int kMaxPyr = 8;

// ------------------------------------------------------------------------------------------------------------------
// 2x2 average of fmap planes: in (planes, h, w) -> out (planes, h/2, w/2), trailing odd row/col dropped
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pool2_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const long total = planes * ho * wo;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int x = (int)(idx % wo);
        const long t = idx / wo;
        const int y = (int)(t % ho);
        const long p = t / ho;
        const float* s = in + (p * h + 2 * y) * w + 2 * x;
        out[idx] = (s[0] + s[1] + s[w] + s[w + 1]) * 0.25f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// magnitudes as bit patterns of non-negative floats (they order like unsigned integers).  Inf / NaN entries do not take part in a
// scale: it comes from the finite values, so a non-finite feature spoils exactly its own row / column of the volume (as in the
// reference's fp32 matmul) and every other entry keeps its full accuracy.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned finite_mag(unsigned bits) {
    const unsigned m = bits & 0x7fffffffu;
    return m >= 0x7f800000u ? 0u : m;
}
// The power of two a pixel's feature vector is divided by before it is split: its largest magnitude lands in [2^14, 2^15) (fp16
// holds up to 65504), so the low terms of all but the very smallest channels stay normal fp16 numbers.  0 for a vector without a
// finite non-zero value.
__device__ __forceinline__ int split_exponent(unsigned absmax_bits) {
    const int e = (int)(absmax_bits >> 23);
    if (e == 0 || e == 255) return 0;
    const int k = e - 127 - 14;
    return k < -110 ? -110 : (k > 110 ? 110 : k);
}

// ------------------------------------------------------------------------------------------------------------------
// per-PIXEL magnitude of a (B, C, n) fp32 feature map: kexp[b][px] = the power of two pixel px's C-vector is divided by before it is
// split (its largest finite magnitude lands in [2^14, 2^15)), and — for the map whose pixels become COLUMNS of the volume — the bit
// pattern of the item's largest finite magnitude (atomicMax; `item_bits` zero on entry, may be null).  A workgroup serves 64 consecutive pixels, each of its four
// waves a quarter of the channels (256 contiguous bytes per load, eight loads in flight per thread).  Scaling every pixel by its OWN power of two makes the accuracy of a
// volume entry relative to |f1_i| |f2_j| (what fp32 matmul gives), not to the item's largest entry: a dim image region next to a
// bright one keeps its 22 bits.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kPixPx = 64, kPixGroups = 4;   // a workgroup: 64 consecutive pixels x 4 channel groups
__global__ void __launch_bounds__(256)
corr_pixmax_kernel(const float* __restrict__ in, int* __restrict__ kexp, unsigned* __restrict__ item_bits, int C, long n) {
    const int b = blockIdx.y;
    const int lp = threadIdx.x & (kPixPx - 1), cg = threadIdx.x >> 6;   // a wave = one channel group: 256 contiguous bytes per load
    const long px = (long)blockIdx.x * kPixPx + lp;
    const bool live = px < n;
    const float* p = in + (long)b * C * n + (live ? px : 0);
    // (float compares, not integer max on the bit patterns: hipcc 7.2's instruction selection crashes on the integer form of this loop)
    auto fin = [](float v) { v = fabsf(v); return v < INFINITY ? v : 0.f; };   // Inf / NaN do not take part
    float f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = 0.f;
    // channels cg, cg + 4, cg + 8, ...: eight independent loads in flight per thread
    int c = cg;
    for (; c + 7 * kPixGroups < C; c += 8 * kPixGroups) {
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = fmaxf(f[u], fin(p[(long)(c + u * kPixGroups) * n]));
    }
    for (; c < C; c += kPixGroups) f[0] = fmaxf(f[0], fin(p[(long)c * n]));
    float m = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), fmaxf(fmaxf(f[4], f[5]), fmaxf(f[6], f[7])));
    __shared__ float part[kPixGroups][kPixPx];
    part[cg][lp] = live ? m : 0.f;
    __syncthreads();
    if (cg != 0) return;   // wave-uniform
    m = fmaxf(fmaxf(part[0][lp], part[1][lp]), fmaxf(part[2][lp], part[3][lp]));
    const unsigned bits = __float_as_uint(m);
    if (live) kexp[(long)b * n + px] = split_exponent(bits);
    if (item_bits == nullptr) return;   // uniform
    unsigned w = bits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w = max(w, (unsigned)__shfl_xor((int)w, o, 64));
    if (lp == 0 && w) atomicMax(item_bits + b, w);
}

// ------------------------------------------------------------------------------------------------------------------
// two-term fp16 split of a feature map: (B, C, n) fp32 -> [b][kc][term][pixel][16 channels] fp16, kc = ceil(C / 16) (channels
// past C are zero): hi = fp16(x'), lo = fp16(x' - hi) with x' = x * 2^-k (exact; k = kexp[b][pixel], corr_pixmax_kernel), both
// round-to-nearest; x' - hi is exact in fp32.  One 16-channel slice of one pixel and one term is 32 contiguous bytes: a GEMM tile's rows of a slice are one
// contiguous run.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSplitPx = 64;
typedef _Float16 __attribute__((ext_vector_type(4))) f16x4_t;

__global__ void __launch_bounds__(256)
corr_split_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, const int* __restrict__ kexp, int C,
                  long n, int KC) {
    __shared__ float tile[16][kSplitPx + 1];
    const long px0 = (long)blockIdx.x * kSplitPx;
    const int kc = blockIdx.y, b = blockIdx.z;
    const int t = threadIdx.x;
    const float down = px0 + (t & 63) < n ? ldexpf(1.0f, -kexp[(long)b * n + px0 + (t & 63)]) : 0.f;   // this thread's pixel
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = (t >> 6) + 4 * e, px = t & 63;
        const int c = kc * 16 + ch;
        tile[ch][px] = (c < C && px0 + px < n) ? in[((long)b * C + c) * n + px0 + px] * down : 0.f;
    }
    __syncthreads();
    const int px = t >> 2, q = t & 3;
    if (px0 + px >= n) return;
    f16x4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = tile[4 * q + e][px];
        hi[e] = (_Float16)x;
        lo[e] = (_Float16)(x - (float)hi[e]);
    }
    const long base = (((long)b * KC + kc) * 2) * n;
    *reinterpret_cast<f16x4_t*>(out + ((base + px0 + px) * 16 + 4 * q)) = hi;
    *reinterpret_cast<f16x4_t*>(out + ((base + n + px0 + px) * 16 + 4 * q)) = lo;
}

// ------------------------------------------------------------------------------------------------------------------
// correlation GEMM on split operands
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}
typedef _Float16 __attribute__((ext_vector_type(8))) f16x8_t;
__device__ __forceinline__ f16x8_t as_f16x8(const u32x4& v) {
    union { u32x4 u; f16x8_t h; } x;
    x.u = v;
    return x.h;
}

struct Gemm3Args {
    const uint16_t* a;   // split fmap1  [B][KC][2][HW][16]
    const uint16_t* b;   // split fmap2 (or a pooled copy of it) [B][KC][2][n][16]
    const int* ka;            // [B][HW] / [B][n]: the power of two every row pixel (fmap1) / column pixel (fmap2 or its pooled copy)
    const int* kb;            // was divided by before the split (corr_pixmax_kernel)
    const unsigned* kb_item;  // [B]: bit pattern of fmap2's largest magnitude per item -> the common exponent columns are brought to
    float* out0;         // POOLED: level 0 (B*HW, H*W); plain: the level (B*HW, n)
    float* out1;         // POOLED: level 1 or null
    float* out2;         // POOLED: level 2 or null
    int B, KC, HW;       // rows
    int H, W, n;         // columns: POOLED: the (H, W) grid, n = H*W; plain: n columns
    int h1, w1, h2, w2;  // POOLED: shapes of levels 1 and 2
    int tiles_m, tiles_r, tiles_c;   // row tiles; column tiles: POOLED (4-row bands) x (32-column strips), plain: tiles_c of 128
    float scale;
    unsigned nblocks;
};

// Workgroup tile: 256 rows (pixels of fmap1) x 128 columns (POOLED: a 4 x 32 patch of fmap2's grid; plain: 128 consecutive
// columns); 4 waves, each 64 rows x all 128 columns = 8 MFMA tiles of 32 x 32 (12 operand reads feed 24 MFMAs).  Per
// 16-channel slice the workgroup needs 16 + 8 KiB of operands (two fp16 terms each); they go from memory STRAIGHT into LDS
// (global_load_lds_dwordx4, no registers, no ds_write pass), one slice ahead, while the matrix pipe works on the current one.
// 2 workgroups = 8 waves per CU (the accumulators take 128 of a wave's registers).
constexpr int kTM = 256, kTN = 128, kGemmThreads = 256;
constexpr int kRowGroup = 4;
constexpr int kGranA = 2 * kTM * 2, kGranB = 2 * kTN * 2;          // 16-byte granules per stage
constexpr int kGemmLds = 2 * (kGranA + kGranB) * 16;               // 48 KiB

// LDS granule of (term, row, half) of an operand tile with ROWS rows.  The LDS-DMA writes lane-linearly, so the layout is
// plain [term][row][half]; the half is flipped on every other group of 8 rows — on the SOURCE address of the DMA and on the
// read address alike — so that the 16 lanes a ds_read_b128 serves together (rows 8 apart share banks) never collide.
template <int ROWS>
__device__ __forceinline__ int slot(int term, int row, int g) { return (term * ROWS + row) * 2 + (g ^ ((row >> 3) & 1)); }

// 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS starting at `lds_dst` (wave-uniform), asynchronously
__device__ __forceinline__ void dma16(const uint16_t* src, u32x4* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, lds_dst, 16, 0, 0);
#endif
}

template <bool POOLED>
__global__ void __launch_bounds__(kGemmThreads, 2)
corr_gemm3_kernel(const Gemm3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const lds0 = reinterpret_cast<u32x4*>(smem_raw);
    auto ldsA = [&](int stage) { return lds0 + stage * (kGranA + kGranB); };
    auto ldsB = [&](int stage) { return lds0 + stage * (kGranA + kGranB) + kGranA; };

    // flat block -> (batch, column tile, row tile); row tiles fastest so neighbours on an XCD reuse the B tile from L2
    const unsigned lb = xcd_contiguous_block(blockIdx.x, g.nblocks);
    // row tiles in groups of kRowGroup: the workgroups resident on an XCD together share kRowGroup A tiles (they stay in its L2
    // for the whole sweep over the column tiles) and stream the B tiles past them
    const int tiles_n = g.tiles_r * g.tiles_c, row_groups = (g.tiles_m + kRowGroup - 1) / kRowGroup;
    const int tm = ((lb / (kRowGroup * tiles_n)) % row_groups) * kRowGroup + lb % kRowGroup;
    const int tn = (lb / kRowGroup) % tiles_n;
    const int b = lb / (kRowGroup * tiles_n * row_groups);
    if (tm >= g.tiles_m) return;   // the last group's padding (workgroup-uniform)
    const int tr = tn / g.tiles_c, tc = tn % g.tiles_c;
    const int i0 = tm * kTM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases, row offsets

    // DMA plan: a wave instruction moves 64 granules = 32 rows x 2 halves of one term.  A: 2 terms x 8 row groups = 16
    // instructions (4 per wave), B: 2 x 4 = 8 (2 per wave).  Rows past the edge re-read row 0 (their accumulators are never
    // stored, and pooled outputs only combine columns that exist).
    const int drow = lane >> 1, dhalf = lane & 1;
    const uint16_t* asrc[4];
    const uint16_t* bsrc[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ins = wave + 4 * e, term = ins >> 3, row = 32 * (ins & 7) + drow;
        const int half = dhalf ^ ((row >> 3) & 1);
        asrc[e] = g.a + ((((long)b * g.KC) * 2 + term) * g.HW + (i0 + row < g.HW ? i0 + row : 0)) * 16 + half * 8;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ins = wave + 4 * e, term = ins >> 2, row = 32 * (ins & 3) + drow;
        const int half = dhalf ^ ((row >> 3) & 1);
        long col;
        bool ok;
        if (POOLED) {
            const int y = 4 * tr + (row >> 5), x = 32 * tc + (row & 31);
            ok = y < g.H && x < g.W;
            col = (long)y * g.W + x;
        } else {
            col = (long)tc * kTN + row;
            ok = col < g.n;
        }
        bsrc[e] = g.b + ((((long)b * g.KC) * 2 + term) * g.n + (ok ? col : 0)) * 16 + half * 8;
    }
    const long a_step = 2L * g.HW * 16, b_step = 2L * g.n * 16;   // one K slice further
    auto stage_in = [&](int kc, int stage) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dma16(asrc[e] + kc * a_step, ldsA(stage) + (wave + 4 * e) * 64);
#pragma unroll
        for (int e = 0; e < 2; ++e) dma16(bsrc[e] + kc * b_step, ldsB(stage) + (wave + 4 * e) * 64);
    };

    // per-PIXEL scaling (exact powers of two): entry (i, j) = acc * 2^(ka_i + kb_j) / sqrt(C).  Columns are first brought to the item's
    // common exponent kref >= every kb_j — acc * cb_j with cb_j = 2^(kb_j - kref) / sqrt(C) <= 1 / sqrt(C): no overflow, and pooled
    // cells add columns on one scale — and v_ldexp_f32 then applies 2^(ka_i + kref) per row (no factor leaves the float range unless
    // the result does).  Same instruction count as one scale per item: a multiply and an ldexp where two multiplies were.
    const int kref = split_exponent(g.kb_item[b]);
    int erow[2][16];
    {
        // registers r = 4 q .. 4 q + 3 of a lane are four CONSECUTIVE rows: one 16-byte load per (h, q) (4-byte aligned: HW is arbitrary);
        // rows past HW read the padding behind the array (kexp_bytes) — their stores are dropped by the range check
        typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        const int* kap = g.ka + (long)b * g.HW + i0 + 64 * wave + 4 * (lane >> 5);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32x4_u v = *reinterpret_cast<const i32x4_u*>(kap + 32 * h + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) erow[h][4 * q + e] = v[e] + kref;
            }
    }
    // (requested HERE, before the K loop: 36 loads whose latency would otherwise stand exposed at the head of the epilogue — measured
    // + 7 % on the kernel when they were issued there)
    float cb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        long col;
        bool in;
        if (POOLED) {
            const int y = 4 * tr + t, x = 32 * tc + (lane & 31);
            in = y < g.H && x < g.W;
            col = (long)y * g.W + x;
        } else {
            col = (long)tc * kTN + 32 * t + (lane & 31);
            in = col < g.n;
        }
        cb[t] = ldexpf(g.scale, (in ? g.kb[(long)b * g.n + col] : kref) - kref);
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.f;

    stage_in(0, 0);
    __syncthreads();   // (hipcc drains the DMA queue in front of a barrier)

    const int kg = lane >> 5, li = lane & 31;
    for (int kc = 0; kc < g.KC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < g.KC) stage_in(kc + 1, cur ^ 1);   // lands while this slice is multiplied
        const u32x4* As = ldsA(cur);
        const u32x4* Bs = ldsB(cur);
        u32x4 af[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int term = 0; term < 2; ++term) af[h][term] = As[slot<kTM>(term, 64 * wave + 32 * h + li, kg)];
        u32x4 bf[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int term = 0; term < 2; ++term) bf[t][term] = Bs[slot<kTN>(term, 32 * t + li, kg)];
        // small cross terms first; the eight tiles rotate, so no MFMA waits on the one before it
#define ALO_CORR_STEP(AT, BT)                                                                                                       \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                               \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                               \
            acc[h][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(af[h][AT]), as_f16x8(bf[t][BT]), acc[h][t], 0, 0, 0);
        ALO_CORR_STEP(1, 0) ALO_CORR_STEP(0, 1) ALO_CORR_STEP(0, 0)
#undef ALO_CORR_STEP
        // keep every MFMA of the slice in front of the barrier: the scheduler otherwise sinks two thirds of them below it, where
        // no DMA is in flight (the next one is issued at the top of the loop), and the staging latency and the matrix work add up
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // the next slice has landed (DMA drained in front of the barrier) and this one is free to overwrite
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): every store
    // instruction writes whole 128-byte lines (32 consecutive columns of two rows).
    // The volume is written once and never read again by this kernel: non-temporal stores keep the stream out of the L2.
    const long rowbase = (long)b * g.HW;
    // undo the operands' power-of-two scaling (exact) together with the 1/sqrt(C): 2^(ka + kb) applied in two halves, so that no
    // factor leaves the float range unless the result does
    if (POOLED) {
        // Every level is addressed through a buffer resource that covers exactly the rows of this tile which exist: the range check
        // drops the rows past HW, and a lane whose column (or pooled cell) does not exist carries an offset past every range — no
        // predicates, no 64-bit address arithmetic: one v_add (row part, a scalar) + the scaling + the store per element.  (The
        // per-element tests and pointer arithmetic of the first version made the epilogue 4 400 instructions per wave; this one runs
        // the whole kernel 9 % faster: tools/exp/corr_gemm_exp.hip.)  The range check looks at the VGPR offset only, so the row part
        // is added there; tile bytes stay below 2^31 (alo_corr_build's grid limit), so kDrop + any row offset neither wraps nor lands
        // inside a range.
        constexpr unsigned kDrop = 0x80000000u;
        const long rows = min((long)(g.HW - i0), (long)kTM);
        const long n1 = (long)g.h1 * g.w1, n2 = (long)g.h2 * g.w2;
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(g.out0 + (rowbase + i0) * g.n, (unsigned)(rows * g.n * 4));
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(g.out1 ? g.out1 + (rowbase + i0) * n1 : nullptr, g.out1 ? (unsigned)(rows * n1 * 4) : 0u);
        const __amdgpu_buffer_rsrc_t r2 = make_rsrc(g.out2 ? g.out2 + (rowbase + i0) * n2 : nullptr, g.out2 ? (unsigned)(rows * n2 * 4) : 0u);
        const unsigned pitch0 = (unsigned)g.n * 4u, pitch1 = (unsigned)n1 * 4u, pitch2 = (unsigned)n2 * 4u;
        const int x = 32 * tc + li, yb = 4 * tr;
        const int x1 = x >> 1, x2 = x >> 2, y2 = yb >> 2;
        unsigned l0[4], l1[2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            l0[t] = (x < g.W && yb + t < g.H) ? ((unsigned)(yb + t) * g.W + x) * 4u + (unsigned)(4 * kg) * pitch0 : kDrop;
#pragma unroll
        for (int p = 0; p < 2; ++p)   // level 1: the even lane of an x pair stores the 2 x 2 mean
            l1[p] = (!(lane & 1) && x1 < g.w1 && (yb >> 1) + p < g.h1) ? ((unsigned)((yb >> 1) + p) * g.w1 + x1) * 4u + (unsigned)(4 * kg) * pitch1
                                                                          : kDrop;
        const unsigned l2 = (!(lane & 3) && x2 < g.w2 && y2 < g.h2) ? ((unsigned)y2 * g.w2 + x2) * 4u + (unsigned)(4 * kg) * pitch2 : kDrop;
        constexpr int kNt = 2;   // aux: non-temporal
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x16 (&acc_h)[4] = acc[h];
            const unsigned wrow = (unsigned)(64 * wave + 32 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = wrow + (unsigned)((r & 3) + 8 * (r >> 2));   // + 4 kg: in the lane part
                float s4 = 0.f;
                const int er = erow[h][r];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float v0 = acc_h[2 * p][r] * cb[2 * p], v1 = acc_h[2 * p + 1][r] * cb[2 * p + 1];   // on the item's common column scale
                    // (keeps the SLP vectoriser from pairing these into v_pk_mul / v_pk_add: the register shuffles around the packed
                    // forms cost more instructions than they save — 1 562 against 1 370 VALU for the kernel)
                    asm volatile("" : "+v"(v0), "+v"(v1));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_ldexpf(v0, er)), r0, l0[2 * p] + row * pitch0, 0, kNt);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_ldexpf(v1, er)), r0, l0[2 * p + 1] + row * pitch0, 0, kNt);
                    // level 1: 2 x 2 mean = this lane's two rows + the same of its x-neighbour (lane ^ 1)
                    float s2 = v0 + v1;
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
                    s4 += s2;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_ldexpf(s2 * 0.25f, er)), r1, l1[p] + row * pitch1, 0, kNt);
                }
                // level 2: 4 x 4 mean = both row pairs + the other half of the quad
                s4 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0x4E, 0xf, 0xf, false));       // quad_perm [2,3,0,1]
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_ldexpf(s4 * 0.0625f, er)), r2, l2 + row * pitch2, 0, kNt);
            }
        }
    } else {
        // plain columns (pyramid levels >= 3 as extra contractions): the same addressing, one resource
        constexpr unsigned kDrop = 0x80000000u;
        const long rows = min((long)(g.HW - i0), (long)kTM);
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(g.out0 + (rowbase + i0) * g.n, (unsigned)(rows * g.n * 4));
        const unsigned pitch0 = (unsigned)g.n * 4u;
        unsigned l0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long col = (long)tc * kTN + 32 * t + li;
            l0[t] = col < g.n ? (unsigned)col * 4u + (unsigned)(4 * kg) * pitch0 : kDrop;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x16 (&acc_h)[4] = acc[h];
            const unsigned wrow = (unsigned)(64 * wave + 32 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned row = wrow + (unsigned)((r & 3) + 8 * (r >> 2));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_ldexpf(acc_h[t][r] * cb[t], erow[h][r])), r0,
                                                          l0[t] + row * pitch0, 0, 2 /* nt */);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// windowed lookup
// ------------------------------------------------------------------------------------------------------------------
constexpr int TQ = 32;  // queries per workgroup: one 128-byte output segment per channel

struct LookupArgs {
    const float* lvl[kMaxPyr];
    int h[kMaxPyr], w[kMaxPyr];
    const float* coords;
    float* out;
    int B, HW, num_levels, tiles_per_batch;
};

// the reference's coordinate round trip (utils.py:8-9, then ATen's unnormalize for align_corners=True)
__device__ __forceinline__ float round_trip(float p, int size) {
#pragma clang fp contract(off)  // the reference rounds the product before anything is subtracted from it
    const float s = (float)(size - 1);
    const float gnorm = 2.0f * p / s - 1.0f;
    return ((gnorm + 1.0f) / 2.0f) * s;
}

// Stage 1 of the lookup, split in two so that a thread can have several strips in flight: `strip_request` turns one (query,
// level, window row) into the strip's position and ISSUES its loads; `strip_finish` interpolates the strip horizontally into
// `hbuf` [ROWS][WIN][TQ] and leaves the vertical fraction / upper staged row of the tap row in `tybuf` / `rowbuf` [WIN][TQ].
//
// A window of WIN x WIN taps spaced one pixel apart touches a (WIN+1)^2 footprint.  Every tap position goes through the
// reference's fp32 round trip on its own, so floor(x_k) may come out as floor(x_0) + k - 1 or + k + 1 when x sits within an ulp
// of an integer; one spare row and column (ROWS = COLS = WIN + 2) lets such taps shift by one.
template <int R>
struct Strip {
    static constexpr int COLS = 2 * R + 3, NV = (COLS + 3) / 4;
    f32x4 raw[NV];      // the strip's taps when mode == 1
    float cx, cy;       // the query's coordinates on this level
    int xb, yb;         // floor of the window's first tap
    int mode;           // 0: nothing to read (all zero), 1: taps are in raw, 2: scalar path (edge strips), -1: query past the end
    const float* src;   // mode 2: the map row the strip lies in
};

template <int R>
__device__ __forceinline__ void strip_request(const LookupArgs& a, int l, int b, int i, int row, Strip<R>& s) {
    constexpr int COLS = Strip<R>::COLS, NV = Strip<R>::NV;
    const int h = a.h[l], w = a.w[l];
    const float inv = 1.0f / (float)(1 << l);
#pragma unroll
    for (int j = 0; j < NV; ++j) s.raw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s.cx = s.cy = 0.f;
    s.xb = s.yb = 0;
    s.src = nullptr;
    s.mode = -1;
    if (i >= a.HW) return;
    s.mode = 0;
    s.cx = a.coords[((long)b * 2 + 0) * a.HW + i] * inv;  // exact: power-of-two scale
    s.cy = a.coords[((long)b * 2 + 1) * a.HW + i] * inv;
    const float iy0 = round_trip(s.cy - (float)R, h);
    const float ix0 = round_trip(s.cx - (float)R, w);
    if (!(fabsf(iy0) < 1e6f && fabsf(ix0) < 1e6f)) {  // NaN too: such queries read as all-zero
        s.mode = -1;
        return;
    }
    s.yb = (int)floorf(iy0);
    s.xb = (int)floorf(ix0);
    const int ry = s.yb + row;
    if (ry >= 0 && ry < h && s.xb + COLS > 0 && s.xb < w) {
        // the strip's COLS taps as 16-byte loads from an arbitrary 4-byte aligned start instead of COLS scalar loads: every
        // load instruction of a wave touches 64 different lines here (one strip per lane, strips 4*h*w bytes apart), so the
        // instruction count is what the L1 is charged for.  Strips that hang off the left edge of the map or would read past
        // the batch item's slab take the scalar path (rare).
        const long e0 = (long)i * h * w + (long)ry * w + s.xb;  // element offset of tap 0 inside the slab
        if (s.xb >= 0 && e0 + 4 * NV <= (long)a.HW * h * w) {
            const float* src = a.lvl[l] + (long)b * a.HW * ((long)h * w) + e0;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                typedef f32x4 __attribute__((aligned(4))) f32x4_u;  // 4-byte aligned vector load
                s.raw[j] = *reinterpret_cast<const f32x4_u*>(src + 4 * j);
            }
            s.mode = 1;
        } else {
            s.src = a.lvl[l] + ((long)b * a.HW + i) * ((long)h * w) + (long)ry * w;
            s.mode = 2;
        }
    }
}

// `dbuf` (optional, same shape as hbuf): hi - lo of every tap = the derivative of the row's interpolated value with respect to the
// tap's x position (the coordinate gradient's horizontal half; zeros outside the map count as values, as in grid_sample's backward)
template <int R>
__device__ __forceinline__ void strip_finish(const LookupArgs& a, int l, int row, int q, const Strip<R>& s, float* hbuf, float* tybuf,
                                             int* rowbuf, float* dbuf = nullptr) {
    constexpr int WIN = 2 * R + 1, COLS = Strip<R>::COLS;
    const int h = a.h[l], w = a.w[l];
    float hv[WIN], dv[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) hv[k] = dv[k] = 0.f;
    float ty = 0.f;
    int trow = row < WIN ? row : 0;
    if (s.mode >= 0) {
        if (row < WIN) {
            const float iy = round_trip(s.cy + (float)(row - R), h);
            const float fy = floorf(iy);
            int d = (int)fy - (s.yb + row);
            d = d < -1 ? -1 : (d > 1 ? 1 : d);
            if (row == 0) d = 0;
            trow = row + d;
            ty = iy - (float)(s.yb + trow);
        }
        if (s.mode > 0) {
            float v[COLS];
            if (s.mode == 1) {
#pragma unroll
                for (int k = 0; k < COLS; ++k) v[k] = (s.xb + k < w) ? s.raw[k / 4][k % 4] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < COLS; ++k) {
                    const int x = s.xb + k;
                    v[k] = (x >= 0 && x < w) ? s.src[x] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float ix = round_trip(s.cx + (float)(k - R), w);
                int d = (int)floorf(ix) - (s.xb + k);
                d = (k == 0) ? 0 : (d < -1 ? -1 : (d > 1 ? 1 : d));
                const float lo = d == 0 ? v[k] : (d > 0 ? v[k + 1] : v[k > 0 ? k - 1 : 0]);
                const float hi = d == 0 ? v[k + 1] : (d > 0 ? v[k + 2] : v[k]);
                const float tx = ix - (float)(s.xb + k + d);
                hv[k] = (1.0f - tx) * lo + tx * hi;
                dv[k] = hi - lo;
            }
        }
    }
    float* dst = hbuf + (row * WIN) * TQ + q;
#pragma unroll
    for (int k = 0; k < WIN; ++k) dst[k * TQ] = hv[k];
    if (dbuf) {
        float* dd = dbuf + (row * WIN) * TQ + q;
#pragma unroll
        for (int k = 0; k < WIN; ++k) dd[k * TQ] = dv[k];
    }
    if (row < WIN) {
        tybuf[row * TQ + q] = ty;
        rowbuf[row * TQ + q] = trow;
    }
}

// Plain lookup: a workgroup serves one (32-query tile, level) with one thread per (query, window-row) strip — every strip read
// of the tile is in flight at once — and 15 KB of LDS, so several workgroups per CU overlap their memory latency.  Stage 2
// interpolates vertically and writes the level's (2r+1)^2 output channels with 128-byte contiguous segments per channel.
template <int R>
constexpr int lookup_threads() { return ((2 * R + 3) * TQ + 63) / 64 * 64; }

template <int R>
__global__ void __launch_bounds__(lookup_threads<R>())
corr_lookup_kernel(const LookupArgs a) {
    constexpr int NT = lookup_threads<R>();
    constexpr int WIN = 2 * R + 1, ROWS = WIN + 2;
    __shared__ float hbuf[ROWS * WIN * TQ];
    __shared__ float tybuf[WIN * TQ];
    __shared__ int rowbuf[WIN * TQ];
    const int b = blockIdx.x / a.tiles_per_batch;
    const int q0 = (blockIdx.x % a.tiles_per_batch) * TQ;
    const int l = blockIdx.y, tid = threadIdx.x;
    if (tid < ROWS * TQ) {
        Strip<R> s;
        strip_request<R>(a, l, b, q0 + tid % TQ, tid / TQ, s);
        strip_finish<R>(a, l, tid / TQ, tid % TQ, s, hbuf, tybuf, rowbuf);
    }
    __syncthreads();
    const int CH = a.num_levels * WIN * WIN;
    for (int o = tid; o < WIN * WIN * TQ; o += NT) {
        const int q = o % TQ, rem = o / TQ;
        const int ax = rem / WIN, cy = rem % WIN;  // first window axis -> x offset, second -> y offset
        const int i = q0 + q;
        if (i < a.HW) {
            const float ty = tybuf[cy * TQ + q];
            const int r0 = rowbuf[cy * TQ + q];
            const float top = hbuf[(r0 * WIN + ax) * TQ + q];
            const float bot = hbuf[((r0 + 1) * WIN + ax) * TQ + q];
            a.out[((long)b * CH + l * WIN * WIN + rem) * a.HW + i] = (1.0f - ty) * top + ty * bot;
        }
    }
}

// Backward of the lookup with respect to the COORDINATES (RAFT detaches them before every lookup, raft.py:186, but the reference's block
// is differentiable there as well: grid_sample's gradient with respect to the grid, chained through the reference's own coordinate
// arithmetic, whose derivative is 1 / 2^l).  Stage 1 is the forward's, with the horizontal difference hi - lo of every tap kept next
// to the interpolated value; stage 2 forms, per tap,  d out / d x = (1 - ty) dtop + ty dbot  and  d out / d y = bot - top,  weighs them
// with the tap's output gradient and sums over the window: a thread keeps the partial sums of ITS query (the thread count is a
// multiple of the tile's 32 queries), twelve partials per query meet in LDS.  One (B, 2, H, W) map per level; the caller adds the levels
// (a fixed order: deterministic).
struct LookupCoordsBwdArgs {
    LookupArgs fwd;       // levels, coords; `out` unused
    const float* gout;    // (B, L * WIN^2, H, W)
    float* gcoords;       // (B, L, 2, H, W)
};

template <int R>
__global__ void __launch_bounds__(lookup_threads<R>())
corr_lookup_coords_bwd_kernel(const LookupCoordsBwdArgs ca) {
    const LookupArgs& a = ca.fwd;
    constexpr int NT = lookup_threads<R>();
    constexpr int WIN = 2 * R + 1, ROWS = WIN + 2;
    static_assert(NT % TQ == 0, "a thread's outputs all belong to one query");
    __shared__ float hbuf[ROWS * WIN * TQ];
    __shared__ float dbuf[ROWS * WIN * TQ];
    __shared__ float tybuf[WIN * TQ];
    __shared__ int rowbuf[WIN * TQ];
    __shared__ float part[2][NT];
    const int b = blockIdx.x / a.tiles_per_batch;
    const int q0 = (blockIdx.x % a.tiles_per_batch) * TQ;
    const int l = blockIdx.y, tid = threadIdx.x;
    if (tid < ROWS * TQ) {
        Strip<R> s;
        strip_request<R>(a, l, b, q0 + tid % TQ, tid / TQ, s);
        strip_finish<R>(a, l, tid / TQ, tid % TQ, s, hbuf, tybuf, rowbuf, dbuf);
    }
    __syncthreads();
    const int CH = a.num_levels * WIN * WIN;
    const int q = tid % TQ, i = q0 + q;
    float gx = 0.f, gy = 0.f;
    if (i < a.HW) {
        for (int rem = tid / TQ; rem < WIN * WIN; rem += NT / TQ) {
            const int ax = rem / WIN, cy = rem % WIN;
            const float ty = tybuf[cy * TQ + q];
            const int r0 = rowbuf[cy * TQ + q];
            const float top = hbuf[(r0 * WIN + ax) * TQ + q], bot = hbuf[((r0 + 1) * WIN + ax) * TQ + q];
            const float dtop = dbuf[(r0 * WIN + ax) * TQ + q], dbot = dbuf[((r0 + 1) * WIN + ax) * TQ + q];
            const float g = ca.gout[((long)b * CH + l * WIN * WIN + rem) * a.HW + i];
            gx += g * ((1.0f - ty) * dtop + ty * dbot);
            gy += g * (bot - top);
        }
    }
    part[0][tid] = gx;
    part[1][tid] = gy;
    __syncthreads();
    if (tid < 2 * TQ) {
        const int c = tid / TQ, qq = tid % TQ, ii = q0 + qq;
        if (ii < a.HW) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NT / TQ; ++k) sum += part[c][k * TQ + qq];
            ca.gcoords[(((long)b * a.num_levels + l) * 2 + c) * a.HW + ii] = sum * (1.0f / (float)(1 << l));
        }
    }
}

// Backward of the lookup with respect to the pyramid (the reference's block is plain torch code — bilinear_sampler = grid_sample —
// so autograd differentiates it, corr.py:29-50; RAFT detaches the coordinates before every lookup, raft.py:186): the exact adjoint of
// the kernel above, ACCUMULATED into gradient maps of the pyramid's own layout.  A query's window lies in its own row of every level,
// so nothing another query touches: a workgroup serves one (32-query tile, level), thread (query, staged row) gathers the row's
// share of the tile's output gradients (vertical weights of the taps that read it), spreads it horizontally over the row's 2r + 3
// staged columns and adds them to the map with plain loads and stores — no atomics, and the 32 lookups of a RAFT forward add into the
// same maps one launch after the other instead of each materialising a dense gradient of the whole volume (what autograd does with
// grid_sample's backward: 4.4 GB per lookup at 720p).  Tap positions, shifts and weights are computed exactly as in strip_finish.
struct LookupBwdArgs {
    float* glvl[kMaxPyr];
    int h[kMaxPyr], w[kMaxPyr];
    const float* coords;
    const float* gout;
    int B, HW, num_levels, tiles_per_batch;
};

template <int R>
__global__ void __launch_bounds__(lookup_threads<R>())
corr_lookup_bwd_kernel(const LookupBwdArgs a) {
    constexpr int NT = lookup_threads<R>();
    constexpr int WIN = 2 * R + 1, ROWS = WIN + 2, COLS = WIN + 2;
    __shared__ float gbuf[WIN * WIN * TQ];   // the tile's output gradients of this level: [ax * WIN + cy][query]
    __shared__ float tybuf[WIN * TQ];
    __shared__ int rowbuf[WIN * TQ];
    const int b = blockIdx.x / a.tiles_per_batch;
    const int q0 = (blockIdx.x % a.tiles_per_batch) * TQ;
    const int l = blockIdx.y, tid = threadIdx.x;
    const int h = a.h[l], w = a.w[l];
    const int CH = a.num_levels * WIN * WIN;
    for (int o = tid; o < WIN * WIN * TQ; o += NT) {
        const int q = o % TQ, rem = o / TQ, i = q0 + q;
        gbuf[o] = i < a.HW ? a.gout[((long)b * CH + l * WIN * WIN + rem) * a.HW + i] : 0.f;
    }
    const int q = tid % TQ, row = tid / TQ, i = q0 + q;
    bool ok = tid < ROWS * TQ && i < a.HW;
    float cx = 0.f, cy = 0.f;
    int xb = 0, yb = 0;
    if (ok) {
        const float inv = 1.0f / (float)(1 << l);
        cx = a.coords[((long)b * 2 + 0) * a.HW + i] * inv;
        cy = a.coords[((long)b * 2 + 1) * a.HW + i] * inv;
        const float iy0 = round_trip(cy - (float)R, h), ix0 = round_trip(cx - (float)R, w);
        ok = fabsf(iy0) < 1e6f && fabsf(ix0) < 1e6f;   // NaN too: such queries read (and therefore spread) nothing
        if (ok) {
            yb = (int)floorf(iy0);
            xb = (int)floorf(ix0);
        }
    }
    if (tid < WIN * TQ) {   // rows 0 .. WIN - 1 double as the window's tap rows: where tap row `row` reads, and with which fraction
        float ty = 0.f;
        int trow = row;
        if (ok) {
            const float iy = round_trip(cy + (float)(row - R), h);
            int d = (int)floorf(iy) - (yb + row);
            d = d < -1 ? -1 : (d > 1 ? 1 : d);
            if (row == 0) d = 0;
            trow = row + d;
            ty = iy - (float)(yb + trow);
        }
        tybuf[row * TQ + q] = ty;
        rowbuf[row * TQ + q] = trow;
    }
    // this thread's map row: requested NOW (interior rows: three 16-byte loads), consumed after the gather below — the round trip
    // to memory runs under the barrier and the two adjoint passes instead of behind them
    constexpr int NV = (COLS + 3) / 4;
    typedef f32x4 __attribute__((aligned(4))) f32x4_u;
    const int ry = yb + row;
    const bool live = ok && ry >= 0 && ry < h;   // rows outside the map were read as zeros
    const bool interior = live && xb >= 0 && xb + 4 * NV <= w;
    float* dst = a.glvl[l] + ((long)b * a.HW + (live ? i : 0)) * ((long)h * w) + (long)(live ? ry : 0) * w;
    f32x4 cur[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) cur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (interior) {
        const f32x4_u* p = reinterpret_cast<const f32x4_u*>(dst + xb);
#pragma unroll
        for (int j = 0; j < NV; ++j) cur[j] = p[j];
    }
    __syncthreads();
    if (!live) return;
    // vertical adjoint: tap row c read staged rows trow_c (weight 1 - ty_c) and trow_c + 1 (weight ty_c)
    float gh[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) gh[k] = 0.f;
    for (int c = 0; c < WIN; ++c) {
        const int r0 = rowbuf[c * TQ + q];
        const float ty = tybuf[c * TQ + q];
        const float wgt = r0 == row ? 1.0f - ty : (r0 + 1 == row ? ty : 0.f);
        if (r0 == row || r0 + 1 == row) {
#pragma unroll
            for (int k = 0; k < WIN; ++k) gh[k] += wgt * gbuf[(k * WIN + c) * TQ + q];
        }
    }
    // horizontal adjoint: tap column k read staged columns k + d and k + d + 1
    float gv[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) gv[j] = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
        const float ix = round_trip(cx + (float)(k - R), w);
        int d = (int)floorf(ix) - (xb + k);
        d = (k == 0) ? 0 : (d < -1 ? -1 : (d > 1 ? 1 : d));
        const float tx = ix - (float)(xb + k + d);
        const float lo = (1.0f - tx) * gh[k], hi = tx * gh[k];
        if (d == 0) {
            gv[k] += lo;
            gv[k + 1] += hi;
        } else if (d > 0) {
            gv[k + 1] += lo;
            gv[k + 2] += hi;
        } else {
            gv[k > 0 ? k - 1 : 0] += lo;
            gv[k] += hi;
        }
    }
    if (interior) {
        // the staged columns (and the spare lanes of the last vector) lie inside this map row, which no other thread touches: 16-byte
        // loads / stores from a 4-byte aligned start, as in the forward — 2 NV memory instructions instead of 2 COLS (the scalar form
        // took 0.53 ms per lookup at 720p, B = 4; the texture path charges per instruction)
        f32x4_u* p = reinterpret_cast<f32x4_u*>(dst + xb);
#pragma unroll
        for (int j = 0; j < COLS; ++j) cur[j / 4][j % 4] += gv[j];
#pragma unroll
        for (int j = 0; j < NV; ++j) p[j] = cur[j];
    } else {
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            const int x = xb + j;
            if (x >= 0 && x < w) dst[x] += gv[j];
        }
    }
}

template <int R>
int launch_lookup(const LookupArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(corr_lookup_kernel<R>, dim3(a.B * a.tiles_per_batch, a.num_levels), dim3(lookup_threads<R>()), 0, stream, a);
    return check_launch("alo_corr_lookup");
}
template <int R>
int launch_lookup_coords_bwd(const LookupCoordsBwdArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(corr_lookup_coords_bwd_kernel<R>, dim3(a.fwd.B * a.fwd.tiles_per_batch, a.fwd.num_levels), dim3(lookup_threads<R>()), 0,
                       stream, a);
    return check_launch("alo_corr_lookup_backward_coords");
}

template <int R>
int launch_lookup_bwd(const LookupBwdArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(corr_lookup_bwd_kernel<R>, dim3(a.B * a.tiles_per_batch, a.num_levels), dim3(lookup_threads<R>()), 0, stream, a);
    return check_launch("alo_corr_lookup_backward");
}

int fill_lookup_args(LookupArgs& a, const float* const* levels, const float* coords, float* out, int B, int H, int W, int radius,
                     int num_levels, const char* what) {
    ALO_REQUIRE(levels && coords && out, ALO_ERR_INVALID_ARGUMENT, "%s: null pointer argument", what);
    ALO_REQUIRE(B > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "%s: dimensions must be positive", what);
    ALO_REQUIRE(radius >= 0 && radius <= 7, ALO_ERR_UNSUPPORTED, "%s: radius must be in [0,7], got %d", what, radius);
    ALO_REQUIRE(num_levels >= 1 && num_levels <= kMaxPyr, ALO_ERR_INVALID_ARGUMENT,
                "%s: num_levels must be in [1,%d], got %d", what, kMaxPyr, num_levels);
    for (int l = 0; l < kMaxPyr; ++l) { a.lvl[l] = nullptr; a.h[l] = a.w[l] = 2; }
    for (int l = 0; l < num_levels; ++l) {
        ALO_REQUIRE(levels[l], ALO_ERR_INVALID_ARGUMENT, "%s: levels[%d] is null", what, l);
        alo_corr_level_shape(H, W, l, &a.h[l], &a.w[l]);
        ALO_REQUIRE(a.h[l] >= 2 && a.w[l] >= 2, ALO_ERR_INVALID_ARGUMENT,
                    "%s: level %d is %dx%d; the reference divides by (size-1) and needs >= 2", what, l, a.h[l], a.w[l]);
        a.lvl[l] = levels[l];
    }
    a.coords = coords;
    a.out = out;
    a.B = B;
    a.HW = H * W;
    a.num_levels = num_levels;
    a.tiles_per_batch = (a.HW + TQ - 1) / TQ;
    return ALO_OK;
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" void alo_corr_level_shape(int H, int W, int level, int* h_out, int* w_out) {
    for (int l = 0; l < level; ++l) { H /= 2; W /= 2; }
    if (h_out) *h_out = H;
    if (w_out) *w_out = W;
}

namespace {
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t split_bytes(int B, int C, long n) { return align256((size_t)B * ((C + 15) / 16) * 2 * n * 16 * sizeof(uint16_t)); }
inline size_t absmax_bytes(int B) { return align256((size_t)B * sizeof(unsigned)); }                 // fmap2's largest magnitude per item
inline size_t kexp_bytes(int B, long n) { return align256(((size_t)B * n + kTM + 4) * sizeof(int)); }   // one power of two per pixel (+ a row tile of padding: the last tile reads whole)
}  // namespace

// Scratch: the split copies of fmap1 and fmap2, their per-pixel powers of two and fmap2's per-item magnitude; for pyramids deeper
// than 3 levels also the 2x2-average chain of fmap2 (fp32) and the split copies + per-pixel powers of two of its levels >= 3.
extern "C" size_t alo_corr_build_workspace_bytes(int B, int C, int H, int W, int num_levels) {
    size_t total = 2 * split_bytes(B, C, (long)H * W) + 2 * kexp_bytes(B, (long)H * W) + absmax_bytes(B);
    for (int l = 1; l < num_levels && num_levels > 3; ++l) {
        int h, w;
        alo_corr_level_shape(H, W, l, &h, &w);
        total += align256((size_t)B * C * h * w * sizeof(float));
        if (l >= 3) total += split_bytes(B, C, (long)h * w) + kexp_bytes(B, (long)h * w);
    }
    return total;
}

extern "C" int alo_corr_build(const float* fmap1, const float* fmap2, float* const* levels, void* workspace,
                              size_t workspace_bytes, int B, int C, int H, int W, int num_levels, void* stream_) {
    ALO_REQUIRE(fmap1 && fmap2 && levels, ALO_ERR_INVALID_ARGUMENT, "alo_corr_build: null pointer argument");
    ALO_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_corr_build: dimensions must be positive (B=%d C=%d H=%d W=%d)", B, C, H, W);
    ALO_REQUIRE(num_levels >= 1 && num_levels <= kMaxPyr, ALO_ERR_INVALID_ARGUMENT,
                "alo_corr_build: num_levels must be in [1,%d], got %d", kMaxPyr, num_levels);
    const size_t need = alo_corr_build_workspace_bytes(B, C, H, W, num_levels);
    ALO_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_corr_build: a 16-byte aligned workspace of %zu bytes is required, %zu given", need, workspace_bytes);
    ALO_REQUIRE((double)H * W * H * W < 2.0e9 * 64, ALO_ERR_UNSUPPORTED, "alo_corr_build: feature grid too large");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const long HW = (long)H * W;
    const int KC = (C + 15) / 16;
    for (int l = 0; l < num_levels; ++l) {
        int h, w;
        alo_corr_level_shape(H, W, l, &h, &w);
        ALO_REQUIRE(h > 0 && w > 0, ALO_ERR_INVALID_ARGUMENT, "alo_corr_build: pyramid level %d is empty (%dx%d grid)", l, H, W);
        ALO_REQUIRE(levels[l], ALO_ERR_INVALID_ARGUMENT, "alo_corr_build: levels[%d] is null", l);
    }

    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint16_t* f1s = reinterpret_cast<uint16_t*>(ws);
    ws += split_bytes(B, C, HW);
    uint16_t* f2s = reinterpret_cast<uint16_t*>(ws);
    ws += split_bytes(B, C, HW);
    int* ka = reinterpret_cast<int*>(ws);               // [B][HW]
    ws += kexp_bytes(B, HW);
    int* kb = reinterpret_cast<int*>(ws);               // [B][HW]
    ws += kexp_bytes(B, HW);
    unsigned* amax_b = reinterpret_cast<unsigned*>(ws);  // [B]
    ws += absmax_bytes(B);
    hipError_t em = hipMemsetAsync(amax_b, 0, (size_t)B * sizeof(unsigned), stream);
    if (em != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_corr_build: memset: %s", hipGetErrorString(em));
    auto pixmax = [&](const float* in, int* kexp, unsigned* item_bits, long n) -> int {
        hipLaunchKernelGGL(corr_pixmax_kernel, dim3((unsigned)((n + kPixPx - 1) / kPixPx), (unsigned)B), dim3(256), 0, stream, in, kexp, item_bits, C, n);
        return check_launch("alo_corr_build(pixmax)");
    };
    if (int rc = pixmax(fmap1, ka, nullptr, HW)) return rc;
    if (int rc = pixmax(fmap2, kb, amax_b, HW)) return rc;
    auto split = [&](const float* in, uint16_t* out, const int* kexp, long n) -> int {
        const dim3 grid((unsigned)((n + kSplitPx - 1) / kSplitPx), (unsigned)KC, (unsigned)B);
        hipLaunchKernelGGL(corr_split_kernel, grid, dim3(256), 0, stream, in, out, kexp, C, n, KC);
        return check_launch("alo_corr_build(split)");
    };
    if (int rc = split(fmap1, f1s, ka, HW)) return rc;
    if (int rc = split(fmap2, f2s, kb, HW)) return rc;

    Gemm3Args g;
    g.a = f1s; g.b = f2s;
    g.ka = ka; g.kb = kb; g.kb_item = amax_b;
    g.B = B; g.KC = KC; g.HW = (int)HW;
    g.H = H; g.W = W; g.n = (int)HW;
    g.tiles_m = (int)((HW + kTM - 1) / kTM);
    g.tiles_r = (H + 3) / 4;
    g.tiles_c = (W + 31) / 32;
    g.scale = 1.0f / sqrtf((float)C);
    g.out0 = levels[0];
    g.out1 = num_levels > 1 ? levels[1] : nullptr;
    g.out2 = num_levels > 2 ? levels[2] : nullptr;
    alo_corr_level_shape(H, W, 1, &g.h1, &g.w1);
    alo_corr_level_shape(H, W, 2, &g.h2, &g.w2);
    long nblocks = (long)((g.tiles_m + kRowGroup - 1) / kRowGroup) * kRowGroup * g.tiles_r * g.tiles_c * B;
    ALO_REQUIRE(nblocks < 0x7fffffffL, ALO_ERR_UNSUPPORTED, "alo_corr_build: grid too large");
    g.nblocks = (unsigned)nblocks;
    static unsigned long long attr_done[2] = {0, 0};   // one bit per device
    {
        hipError_t e1 = ensure_dynamic_lds(reinterpret_cast<const void*>(corr_gemm3_kernel<true>), kGemmLds, &attr_done[0]);
        hipError_t e2 = ensure_dynamic_lds(reinterpret_cast<const void*>(corr_gemm3_kernel<false>), kGemmLds, &attr_done[1]);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_corr_build: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    hipLaunchKernelGGL(corr_gemm3_kernel<true>, dim3(g.nblocks), dim3(kGemmThreads), kGemmLds, stream, g);
    if (int rc = check_launch("alo_corr_build(gemm)")) return rc;

    // levels >= 3: the same contraction against the 2x2-average chain of fmap2
    const float* prev = fmap2;
    int ph = H, pw = W;
    for (int l = 1; l < num_levels && num_levels > 3; ++l) {
        int h, w;
        alo_corr_level_shape(H, W, l, &h, &w);
        float* pooled = reinterpret_cast<float*>(ws);
        ws += align256((size_t)B * C * h * w * sizeof(float));
        const long planes = (long)B * C, total = planes * h * w;
        const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        hipLaunchKernelGGL(pool2_kernel, dim3(blocks), dim3(256), 0, stream, prev, pooled, planes, ph, pw);
        if (int rc = check_launch("alo_corr_build(pool)")) return rc;
        prev = pooled; ph = h; pw = w;
        if (l < 3) continue;
        const long n = (long)h * w;
        uint16_t* ps = reinterpret_cast<uint16_t*>(ws);
        ws += split_bytes(B, C, n);
        int* kp = reinterpret_cast<int*>(ws);
        ws += kexp_bytes(B, n);
        if (int rc = pixmax(pooled, kp, nullptr, n)) return rc;   // (a 2x2 mean is no larger than fmap2's largest entry: kb_item still bounds it)
        if (int rc = split(pooled, ps, kp, n)) return rc;
        Gemm3Args e = g;
        e.b = ps; e.kb = kp; e.n = (int)n;
        e.out0 = levels[l]; e.out1 = e.out2 = nullptr;
        e.tiles_r = 1;
        e.tiles_c = (int)((n + kTN - 1) / kTN);
        e.nblocks = (unsigned)((long)((e.tiles_m + kRowGroup - 1) / kRowGroup) * kRowGroup * e.tiles_c * B);
        hipLaunchKernelGGL(corr_gemm3_kernel<false>, dim3(e.nblocks), dim3(kGemmThreads), kGemmLds, stream, e);
        if (int rc = check_launch("alo_corr_build(gemm, coarse level)")) return rc;
    }
    return ALO_OK;
}

extern "C" int alo_corr_lookup(const float* const* levels, const float* coords, float* out, int B, int H, int W,
                               int radius, int num_levels, void* stream_) {
    LookupArgs a;
    if (int rc = fill_lookup_args(a, levels, coords, out, B, H, W, radius, num_levels, "alo_corr_lookup")) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    switch (radius) {
        case 0: return launch_lookup<0>(a, stream);
        case 1: return launch_lookup<1>(a, stream);
        case 2: return launch_lookup<2>(a, stream);
        case 3: return launch_lookup<3>(a, stream);
        case 4: return launch_lookup<4>(a, stream);
        case 5: return launch_lookup<5>(a, stream);
        case 6: return launch_lookup<6>(a, stream);
        default: return launch_lookup<7>(a, stream);
    }
}

extern "C" int alo_corr_lookup_backward(float* const* grad_levels, const float* coords, const float* grad_out, int B, int H, int W,
                                        int radius, int num_levels, void* stream_) {
    const char* what = "alo_corr_lookup_backward";
    ALO_REQUIRE(grad_levels && coords && grad_out, ALO_ERR_INVALID_ARGUMENT, "%s: null pointer argument", what);
    ALO_REQUIRE(B > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "%s: dimensions must be positive", what);
    ALO_REQUIRE(radius >= 0 && radius <= 7, ALO_ERR_UNSUPPORTED, "%s: radius must be in [0,7], got %d", what, radius);
    ALO_REQUIRE(num_levels >= 1 && num_levels <= kMaxPyr, ALO_ERR_INVALID_ARGUMENT, "%s: num_levels must be in [1,%d], got %d", what,
                kMaxPyr, num_levels);
    LookupBwdArgs a;
    for (int l = 0; l < kMaxPyr; ++l) { a.glvl[l] = nullptr; a.h[l] = a.w[l] = 2; }
    for (int l = 0; l < num_levels; ++l) {
        ALO_REQUIRE(grad_levels[l], ALO_ERR_INVALID_ARGUMENT, "%s: grad_levels[%d] is null", what, l);
        alo_corr_level_shape(H, W, l, &a.h[l], &a.w[l]);
        ALO_REQUIRE(a.h[l] >= 2 && a.w[l] >= 2, ALO_ERR_INVALID_ARGUMENT,
                    "%s: level %d is %dx%d; the reference divides by (size-1) and needs >= 2", what, l, a.h[l], a.w[l]);
        a.glvl[l] = grad_levels[l];
    }
    a.coords = coords;
    a.gout = grad_out;
    a.B = B;
    a.HW = H * W;
    a.num_levels = num_levels;
    a.tiles_per_batch = (a.HW + TQ - 1) / TQ;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    switch (radius) {
        case 0: return launch_lookup_bwd<0>(a, stream);
        case 1: return launch_lookup_bwd<1>(a, stream);
        case 2: return launch_lookup_bwd<2>(a, stream);
        case 3: return launch_lookup_bwd<3>(a, stream);
        case 4: return launch_lookup_bwd<4>(a, stream);
        case 5: return launch_lookup_bwd<5>(a, stream);
        case 6: return launch_lookup_bwd<6>(a, stream);
        default: return launch_lookup_bwd<7>(a, stream);
    }
}

extern "C" int alo_corr_lookup_backward_coords(const float* const* levels, const float* coords, const float* grad_out,
                                               float* grad_coords_levels, int B, int H, int W, int radius, int num_levels, void* stream_) {
    const char* what = "alo_corr_lookup_backward_coords";
    ALO_REQUIRE(grad_out, ALO_ERR_INVALID_ARGUMENT, "%s: null pointer argument", what);
    LookupCoordsBwdArgs ca;
    if (int rc = fill_lookup_args(ca.fwd, levels, coords, grad_coords_levels, B, H, W, radius, num_levels, what)) return rc;
    ca.gout = grad_out;
    ca.gcoords = grad_coords_levels;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    switch (radius) {
        case 0: return launch_lookup_coords_bwd<0>(ca, stream);
        case 1: return launch_lookup_coords_bwd<1>(ca, stream);
        case 2: return launch_lookup_coords_bwd<2>(ca, stream);
        case 3: return launch_lookup_coords_bwd<3>(ca, stream);
        case 4: return launch_lookup_coords_bwd<4>(ca, stream);
        case 5: return launch_lookup_coords_bwd<5>(ca, stream);
        case 6: return launch_lookup_coords_bwd<6>(ca, stream);
        default: return launch_lookup_coords_bwd<7>(ca, stream);
    }
}
