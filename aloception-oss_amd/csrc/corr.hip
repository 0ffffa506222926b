// RAFT all-pairs correlation pyramid + windowed lookup for gfx950 (MI355X).
//
// Reference semantics: alonet/raft/corr.py:13-60 (volume = fmap1^T . fmap2 / sqrt(C); 3x avg_pool2d; 9x9 bilinear
// window per level) and alonet/raft/utils/utils.py:5-19 (pixel -> [-1,1] -> grid_sample(align_corners=True)).
//
// Build: average pooling over the (h2, w2) axes of the volume is linear, so
//        level_l[b,i,:] = <fmap1[b,:,i], pool^l(fmap2)[b,:,:]> / sqrt(C):
// every pyramid level is the same dense contraction against a (tiny) pooled copy of fmap2.  One launch covers all
// levels; the contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so the volume
// keeps the reference's fp32 accuracy) and each level is written exactly once — the 3.3 GB level-0 volume is never
// re-read to make the coarser levels, and the 1/sqrt(C) scale is folded into the epilogue.
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
// all-levels correlation GEMM on fp32 MFMA
// ------------------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;

struct GemmLevel {
    const float* b;  // (B, C, n) pooled fmap2 of this level
    float* out;      // (B*HW, n)
    int n;           // h_l * w_l
    int tile0;       // first column-tile index of this level in the flat tile list
};
struct GemmArgs {
    const float* a;  // fmap1 (B, C, HW)
    int B, C, HW;
    int tiles_m;     // ceil(HW / BM)
    int tiles_n;     // sum over levels of ceil(n_l / BN)
    int num_levels;
    float scale;     // 1 / sqrt(C)
    unsigned nblocks;
    GemmLevel lvl[kMaxPyr];
};

// Stage a BK x 128 panel (k-major, 128 contiguous columns) into registers: 4 x float4 per thread.
template <bool ALIGNED>
__device__ __forceinline__ void panel_load(const float* __restrict__ src, int ld, int rows_valid, int cols_valid,
                                           int tid, f32x4 (&r)[4]) {
    const int c4 = (tid & 31) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (tid >> 5) + 8 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < rows_valid) {
            const float* p = src + (long)k * ld + c4;
            if (ALIGNED) {
                if (c4 < cols_valid) v = *reinterpret_cast<const f32x4*>(p);  // ld % 4 == 0: all four in or out
            } else {
                if (c4 + 0 < cols_valid) v.x = p[0];
                if (c4 + 1 < cols_valid) v.y = p[1];
                if (c4 + 2 < cols_valid) v.z = p[2];
                if (c4 + 3 < cols_valid) v.w = p[3];
            }
        }
        r[j] = v;
    }
}
__device__ __forceinline__ void panel_store(float* __restrict__ dst, int tid, const f32x4 (&r)[4]) {
    const int c4 = (tid & 31) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = (tid >> 5) + 8 * j;
        *reinterpret_cast<f32x4*>(dst + k * BM + c4) = r[j];
    }
}

template <bool ALIGNED>
__global__ void __launch_bounds__(256, 2)
corr_gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BK * BM];  // [buffer][A|B][k][col]  64 KiB

    // flat block -> (batch, column tile, row tile); row tiles fastest so neighbours on an XCD reuse the B panel from L2
    const unsigned lb = xcd_contiguous_block(blockIdx.x, g.nblocks);
    const int tm = lb % g.tiles_m;
    const int tn_flat = (lb / g.tiles_m) % g.tiles_n;
    const int b = lb / (g.tiles_m * g.tiles_n);
    int li = 0;
#pragma unroll
    for (int l = 1; l < kMaxPyr; ++l)
        if (l < g.num_levels && tn_flat >= g.lvl[l].tile0) li = l;
    const GemmLevel L = g.lvl[li];
    const int i0 = tm * BM;
    const int j0 = (tn_flat - L.tile0) * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const float* Ap = g.a + (long)b * g.C * g.HW + i0;
    const float* Bp = L.b + (long)b * g.C * L.n + j0;
    const int a_cols = g.HW - i0, b_cols = L.n - j0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    const int nk = (g.C + BK - 1) / BK;
    panel_load<ALIGNED>(Ap, g.HW, g.C, a_cols, tid, ra);
    panel_load<ALIGNED>(Bp, L.n, g.C, b_cols, tid, rb);
    panel_store(lds[0][0], tid, ra);
    panel_store(lds[0][1], tid, rb);
    __syncthreads();

    // operand fetch: lane holds A[i = lane & 31][k = lane >> 5]; the wave's two 32-row tiles are interleaved
    // (row 2r + t of the 64-row strip belongs to tile t) so ONE 8-byte LDS read feeds both tiles, conflict-free.
    const int a_off = (lane >> 5) * BM + wm * 64 + 2 * (lane & 31);
    const int b_off = (lane >> 5) * BM + wn * 64 + 2 * (lane & 31);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int k0 = (kt + 1) * BK;
            panel_load<ALIGNED>(Ap + (long)k0 * g.HW, g.HW, g.C - k0, a_cols, tid, ra);
            panel_load<ALIGNED>(Bp + (long)k0 * L.n, L.n, g.C - k0, b_cols, tid, rb);
        }
        const float* As = lds[cur][0];
        const float* Bs = lds[cur][1];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float2 av = *reinterpret_cast<const float2*>(As + kk * 2 * BM + a_off);
            const float2 bv = *reinterpret_cast<const float2*>(Bs + kk * 2 * BM + b_off);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            panel_store(lds[cur ^ 1][0], tid, ra);
            panel_store(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* outp = L.out + ((long)b * g.HW) * L.n;
    const int jc = j0 + wn * 64 + 2 * (lane & 31);
    const bool pair_ok = ((L.n & 1) == 0);  // even row length: (i*n + jc) is even -> 8-byte aligned float2 stores
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int i = i0 + wm * 64 + 2 * row + ti;
            if (i < g.HW) {
                float* p = outp + (long)i * L.n + jc;
                const float v0 = acc[ti][0][r] * g.scale, v1 = acc[ti][1][r] * g.scale;
                // the volume is written once and not read again by this kernel: non-temporal stores keep the 4.4 GB stream
                // from being read for ownership / parked in the L2
                if (pair_ok && jc + 1 < L.n) {
                    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
                    __builtin_nontemporal_store(f32x2_t{v0, v1}, reinterpret_cast<f32x2_t*>(p));
                } else {
                    if (jc < L.n) __builtin_nontemporal_store(v0, p);
                    if (jc + 1 < L.n) __builtin_nontemporal_store(v1, p + 1);
                }
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

template <int R>
__global__ void __launch_bounds__(256)
corr_lookup_kernel(const LookupArgs a) {
    // A window of WIN x WIN taps spaced one pixel apart touches a (WIN+1)^2 footprint.  Every tap position goes through
    // the reference's fp32 round trip on its own, so floor(x_k) may come out as floor(x_0) + k - 1 or + k + 1 when x sits
    // within an ulp of an integer; one spare row and column (ROWS = COLS = WIN + 2) lets such taps shift by one.
    constexpr int WIN = 2 * R + 1, ROWS = WIN + 2, COLS = WIN + 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hbuf = sm;                                            // [L][ROWS][WIN][TQ] horizontally interpolated rows
    float* tybuf = sm + a.num_levels * ROWS * WIN * TQ;          // [L][WIN][TQ]       vertical fraction per tap row
    int* rowbuf = reinterpret_cast<int*>(tybuf + a.num_levels * WIN * TQ);  // [L][WIN][TQ] upper staged row per tap row

    const int b = blockIdx.x / a.tiles_per_batch;
    const int q0 = (blockIdx.x % a.tiles_per_batch) * TQ;
    const int tid = threadIdx.x;

    const int items = a.num_levels * ROWS * TQ;
    for (int item = tid; item < items; item += 256) {
        const int q = item % TQ;
        const int row = (item / TQ) % ROWS;
        const int l = item / (TQ * ROWS);
        const int i = q0 + q;
        float hv[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) hv[k] = 0.f;
        float ty = 0.f;
        int trow = row < WIN ? row : 0;
        if (i < a.HW) {
            const int h = a.h[l], w = a.w[l];
            const float inv = 1.0f / (float)(1 << l);
            const float cx = a.coords[((long)b * 2 + 0) * a.HW + i] * inv;  // exact: power-of-two scale
            const float cy = a.coords[((long)b * 2 + 1) * a.HW + i] * inv;
            const float iy0 = round_trip(cy - (float)R, h);
            const float ix0 = round_trip(cx - (float)R, w);
            if (fabsf(iy0) < 1e6f && fabsf(ix0) < 1e6f) {  // false for NaN too: such queries read as all-zero
                const int yb = (int)floorf(iy0), xb = (int)floorf(ix0);
                const int ry = yb + row;
                if (row < WIN) {
                    const float iy = round_trip(cy + (float)(row - R), h);
                    const float fy = floorf(iy);
                    int d = (int)fy - (yb + row);
                    d = d < -1 ? -1 : (d > 1 ? 1 : d);
                    if (row == 0) d = 0;
                    trow = row + d;
                    ty = iy - (float)(yb + trow);
                }
                if (ry >= 0 && ry < h && xb + COLS > 0 && xb < w) {
                    // the strip's COLS taps as 16-byte loads from an arbitrary 4-byte aligned start instead of COLS scalar
                    // loads: every load instruction of a wave touches 64 different lines here (one strip per lane, strips
                    // 4*h*w bytes apart), so the instruction count is what the L1 is charged for.  Strips that hang off the
                    // left edge of the map or would read past the batch item's slab take the scalar path (rare).
                    const long e0 = (long)i * h * w + (long)ry * w + xb;  // element offset of tap 0 inside the slab
                    float v[COLS];
                    constexpr int NV = (COLS + 3) / 4;
                    if (xb >= 0 && e0 + 4 * NV <= (long)a.HW * h * w) {
                        const float* src = a.lvl[l] + (long)b * a.HW * ((long)h * w) + e0;
                        f32x4 raw[NV];
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            typedef f32x4 __attribute__((aligned(4))) f32x4_u;  // 4-byte aligned vector load
                            raw[j] = *reinterpret_cast<const f32x4_u*>(src + 4 * j);
                        }
#pragma unroll
                        for (int k = 0; k < COLS; ++k) v[k] = (xb + k < w) ? raw[k / 4][k % 4] : 0.f;
                    } else {
                        const float* src = a.lvl[l] + ((long)b * a.HW + i) * ((long)h * w) + (long)ry * w;
#pragma unroll
                        for (int k = 0; k < COLS; ++k) {
                            const int x = xb + k;
                            v[k] = (x >= 0 && x < w) ? src[x] : 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < WIN; ++k) {
                        const float ix = round_trip(cx + (float)(k - R), w);
                        int d = (int)floorf(ix) - (xb + k);
                        d = (k == 0) ? 0 : (d < -1 ? -1 : (d > 1 ? 1 : d));
                        const float lo = d == 0 ? v[k] : (d > 0 ? v[k + 1] : v[k > 0 ? k - 1 : 0]);
                        const float hi = d == 0 ? v[k + 1] : (d > 0 ? v[k + 2] : v[k]);
                        const float tx = ix - (float)(xb + k + d);
                        hv[k] = (1.0f - tx) * lo + tx * hi;
                    }
                }
            }
        }
        float* dst = hbuf + ((l * ROWS + row) * WIN) * TQ + q;
#pragma unroll
        for (int k = 0; k < WIN; ++k) dst[k * TQ] = hv[k];
        if (row < WIN) {
            tybuf[(l * WIN + row) * TQ + q] = ty;
            rowbuf[(l * WIN + row) * TQ + q] = trow;
        }
    }
    __syncthreads();

    const int CH = a.num_levels * WIN * WIN;
    for (int o = tid; o < CH * TQ; o += 256) {
        const int q = o % TQ, ch = o / TQ;
        const int l = ch / (WIN * WIN), rem = ch % (WIN * WIN);
        const int ax = rem / WIN, cy = rem % WIN;  // first window axis -> x offset, second -> y offset
        const int i = q0 + q;
        if (i < a.HW) {
            const float ty = tybuf[(l * WIN + cy) * TQ + q];
            const int r0 = rowbuf[(l * WIN + cy) * TQ + q];
            const float top = hbuf[((l * ROWS + r0) * WIN + ax) * TQ + q];
            const float bot = hbuf[((l * ROWS + r0 + 1) * WIN + ax) * TQ + q];
            a.out[((long)b * CH + ch) * a.HW + i] = (1.0f - ty) * top + ty * bot;
        }
    }
}

template <int R>
int launch_lookup(const LookupArgs& a, hipStream_t stream) {
    constexpr int WIN = 2 * R + 1, ROWS = WIN + 2;
    const size_t lds = (size_t)a.num_levels * (ROWS * WIN + 2 * WIN) * TQ * sizeof(float);
    if (lds > 160 * 1024) return fail(ALO_ERR_UNSUPPORTED, "alo_corr_lookup: window too large for LDS");
    auto kern = corr_lookup_kernel<R>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_corr_lookup: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_per_batch), dim3(256), lds, stream, a);
    return check_launch("alo_corr_lookup");
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" void alo_corr_level_shape(int H, int W, int level, int* h_out, int* w_out) {
    for (int l = 0; l < level; ++l) { H /= 2; W /= 2; }
    if (h_out) *h_out = H;
    if (w_out) *w_out = W;
}

extern "C" size_t alo_corr_build_workspace_bytes(int B, int C, int H, int W, int num_levels) {
    size_t total = 0;
    for (int l = 1; l < num_levels; ++l) {
        int h, w;
        alo_corr_level_shape(H, W, l, &h, &w);
        total += (size_t)B * C * h * w * sizeof(float);
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
    ALO_REQUIRE(workspace_bytes >= need && (need == 0 || workspace), ALO_ERR_INVALID_ARGUMENT,
                "alo_corr_build: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    ALO_REQUIRE((double)H * W * H * W < 2.0e9 * 64, ALO_ERR_UNSUPPORTED, "alo_corr_build: feature grid too large");
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    GemmArgs g;
    g.a = fmap1;
    g.B = B; g.C = C; g.HW = H * W;
    g.tiles_m = (g.HW + BM - 1) / BM;
    g.num_levels = num_levels;
    g.scale = 1.0f / sqrtf((float)C);
    int tiles = 0;
    const float* prev = fmap2;
    int ph = H, pw = W;
    float* ws = static_cast<float*>(workspace);
    bool aligned = (g.HW % 4 == 0) && (((uintptr_t)fmap1 | (uintptr_t)fmap2) & 15) == 0;
    for (int l = 0; l < num_levels; ++l) {
        int h, w;
        alo_corr_level_shape(H, W, l, &h, &w);
        ALO_REQUIRE(h > 0 && w > 0, ALO_ERR_INVALID_ARGUMENT, "alo_corr_build: pyramid level %d is empty (%dx%d grid)", l, H, W);
        ALO_REQUIRE(levels[l], ALO_ERR_INVALID_ARGUMENT, "alo_corr_build: levels[%d] is null", l);
        const float* bl = fmap2;
        if (l > 0) {
            const long planes = (long)B * C;
            const long total = planes * h * w;
            const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
            hipLaunchKernelGGL(pool2_kernel, dim3(blocks), dim3(256), 0, stream, prev, ws, planes, ph, pw);
            if (int rc = check_launch("alo_corr_build(pool)")) return rc;
            bl = ws;
            ws += total;
        }
        g.lvl[l].b = bl;
        g.lvl[l].out = levels[l];
        g.lvl[l].n = h * w;
        g.lvl[l].tile0 = tiles;
        tiles += (h * w + BN - 1) / BN;
        aligned = aligned && ((h * w) % 4 == 0) && (((uintptr_t)bl) & 15) == 0;
        prev = bl;
        ph = h; pw = w;
    }
    for (int l = num_levels; l < kMaxPyr; ++l) g.lvl[l] = GemmLevel{nullptr, nullptr, 0, 0x7fffffff};
    g.tiles_n = tiles;
    const long nblocks = (long)g.tiles_m * tiles * B;
    ALO_REQUIRE(nblocks < 0x7fffffffL, ALO_ERR_UNSUPPORTED, "alo_corr_build: grid too large");
    g.nblocks = (unsigned)nblocks;
    if (aligned)
        hipLaunchKernelGGL(corr_gemm_kernel<true>, dim3(g.nblocks), dim3(256), 0, stream, g);
    else
        hipLaunchKernelGGL(corr_gemm_kernel<false>, dim3(g.nblocks), dim3(256), 0, stream, g);
    return check_launch("alo_corr_build(gemm)");
}

extern "C" int alo_corr_lookup(const float* const* levels, const float* coords, float* out, int B, int H, int W,
                               int radius, int num_levels, void* stream_) {
    ALO_REQUIRE(levels && coords && out, ALO_ERR_INVALID_ARGUMENT, "alo_corr_lookup: null pointer argument");
    ALO_REQUIRE(B > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_corr_lookup: dimensions must be positive");
    ALO_REQUIRE(radius >= 0 && radius <= 7, ALO_ERR_UNSUPPORTED, "alo_corr_lookup: radius must be in [0,7], got %d", radius);
    ALO_REQUIRE(num_levels >= 1 && num_levels <= kMaxPyr, ALO_ERR_INVALID_ARGUMENT,
                "alo_corr_lookup: num_levels must be in [1,%d], got %d", kMaxPyr, num_levels);
    LookupArgs a;
    for (int l = 0; l < kMaxPyr; ++l) { a.lvl[l] = nullptr; a.h[l] = a.w[l] = 2; }
    for (int l = 0; l < num_levels; ++l) {
        ALO_REQUIRE(levels[l], ALO_ERR_INVALID_ARGUMENT, "alo_corr_lookup: levels[%d] is null", l);
        alo_corr_level_shape(H, W, l, &a.h[l], &a.w[l]);
        ALO_REQUIRE(a.h[l] >= 2 && a.w[l] >= 2, ALO_ERR_INVALID_ARGUMENT,
                    "alo_corr_lookup: level %d is %dx%d; the reference divides by (size-1) and needs >= 2", l, a.h[l], a.w[l]);
        a.lvl[l] = levels[l];
    }
    a.coords = coords;
    a.out = out;
    a.B = B;
    a.HW = H * W;
    a.num_levels = num_levels;
    a.tiles_per_batch = (a.HW + TQ - 1) / TQ;
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
