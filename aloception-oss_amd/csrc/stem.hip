// ResNet stem in one kernel: 7x7 / stride 2 / padding 3 convolution of a 3-channel image, folded batch-norm bias, ReLU and the
// 3x3 / stride 2 / padding 1 max-pool (alonet/detr/backbone.py:19-47 + torchvision ResNet.conv1/bn1/relu/maxpool).
//
// Stock, this is a 147-tap convolution writing 64 channels at half resolution (273 MB at 8 x 800 x 1333), a bias + ReLU pass over
// it and a pooling pass reading it back.  Here the half-resolution map never exists: a workgroup computes the 17 x 15 convolution
// outputs behind 8 x 7 pooled pixels as an implicit GEMM on v_mfma_f32_32x32x16_bf16 (255 pixels = 8 MFMA row tiles, 64
// channels = 2 column tiles, K = 7 tap rows x 24 = 7 taps x 3 channels + 3 zero columns, so that a k-group of 8 never crosses a
// tap row and every A fragment is 16 contiguous bytes of the staged image), keeps them in LDS as bf16 and pools from there.
// The packed weights (22.5 KB) live in registers for the life of the (persistent) workgroup.
#include "common.hpp"

namespace alo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

constexpr int kPoolRows = 8, kPoolCols = 7;                          // pooled pixels per tile
constexpr int kConvRows = 2 * kPoolRows + 1, kConvCols = 2 * kPoolCols + 1;   // 17 x 15 convolution outputs
constexpr int kConvPix = kConvRows * kConvCols;                      // 255
constexpr int kInRows = 2 * kConvRows + 5 + 1;                       // 39 input rows + 1 read (times zero weights) by k-group 21
constexpr int kInCols = 2 * kConvCols + 5;                           // 35 input pixels per row
constexpr int kInElems = kInCols * 3;                                // 105 staged elements per row
constexpr int kInStride = 224;                                       // LDS bytes per staged row (112 elements)
constexpr int kConvStride = 64 * 2 + 16;                             // LDS bytes per convolution output pixel
constexpr int kKsteps = 11;                                          // 21 k-groups of 8 (+1 of zeros)
constexpr int kStemThreads = 256;
constexpr int kStemLds = kInRows * kInStride + kConvPix * kConvStride + 64 * 4;
static_assert(kConvPix <= 8 * 32, "8 MFMA row tiles");

struct StemDims {
    int N, H, W, Hc, Wc, Hp, Wp;
    long sN, sC, sH, sW;   // element strides of the (N, 3, H, W) input
    int tiles_y, tiles_x;
};

__device__ __forceinline__ unsigned max_u16x2(unsigned a, unsigned b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    union { unsigned u; u16x2 v; } x, y, z;
    x.u = a; y.u = b;
    z.v = __builtin_elementwise_max(x.v, y.v);
    return z.u;
}

__global__ void __launch_bounds__(kStemThreads, 2)
stem_conv_pool_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ bias,
                      bf16_t* __restrict__ Y, const StemDims dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const in_lds = smem;
    unsigned char* const conv_lds = smem + kInRows * kInStride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nl = lane & 31, kg = lane >> 5;

    // the packed (64, 176) weight matrix, B-fragment order [2 column tiles][11 k-steps][64 lanes][8]: resident in registers
    u32x4 wreg[kKsteps][2];
#pragma unroll
    for (int s = 0; s < kKsteps; ++s)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wreg[s][ct] = *reinterpret_cast<const u32x4*>(Wp + ((size_t)(ct * kKsteps + s) * 64 + lane) * 8);
    // bias table in LDS, [kg][column tile][16 accumulator registers]: register r of lane (nl, kg) is channel
    // 32 ct + (r & 3) + 8 (r >> 2) + 4 kg — the accumulators START from the bias
    float* const bias_lds = reinterpret_cast<float*>(smem + kInRows * kInStride + kConvPix * kConvStride);
    if (tid < 64) {
        const int r = tid & 15, ct = (tid >> 4) & 1, k2 = tid >> 5;
        bias_lds[tid] = bias ? bf16_to_f32(bias[32 * ct + (r & 3) + 8 * (r >> 2) + 4 * k2].bits) : 0.f;
    }

    // elements 105..111 of every staged row are read (against zero weights) but never staged: make them finite once
    for (int i = tid; i < kInRows * (kInStride / 2 - kInElems); i += kStemThreads) {
        const int row = i / (kInStride / 2 - kInElems), e = kInElems + i % (kInStride / 2 - kInElems);
        *reinterpret_cast<uint16_t*>(in_lds + row * kInStride + e * 2) = 0;
    }

    const int tiles_per_image = dm.tiles_y * dm.tiles_x;
    const int ntiles = tiles_per_image * dm.N;

    // 40 rows x 35 pixels x 3 channels of the image per tile, requested one tile AHEAD (register staged) and written to LDS
    // interleaved [row][x * 3 + c], zero outside the image (the convolution's padding)
    constexpr int kStageIters = (kInRows * kInElems + kStemThreads - 1) / kStemThreads;
    uint16_t sv[kStageIters];
    auto load_tile = [&](int tl) {
        const int n = tl / tiles_per_image, trem = tl - n * tiles_per_image;
        const int ty = trem / dm.tiles_x, tx = trem - ty * dm.tiles_x;
        const int iy0 = 4 * ty * kPoolRows - 5, ix0 = 4 * tx * kPoolCols - 5;
        const bf16_t* xn = X + (size_t)n * dm.sN;
#pragma unroll
        for (int it = 0; it < kStageIters; ++it) {
            const int i = tid + it * kStemThreads;
            const int row = i / kInElems, e = i - row * kInElems, xc = e / 3, c = e - 3 * xc;
            const int iy = iy0 + row, ix = ix0 + xc;
            const bool ok = i < kInRows * kInElems && iy >= 0 && iy < dm.H && ix >= 0 && ix < dm.W;
            sv[it] = ok ? xn[c * dm.sC + iy * dm.sH + ix * dm.sW].bits : (uint16_t)0;
        }
    };
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_image, trem = tile - n * tiles_per_image;
        const int ty = trem / dm.tiles_x, tx = trem - ty * dm.tiles_x;
        const int py0 = ty * kPoolRows, px0 = tx * kPoolCols;          // first pooled pixel
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;                // first convolution output

#pragma unroll
        for (int it = 0; it < kStageIters; ++it) {
            const int i = tid + it * kStemThreads;
            const int row = i / kInElems, e = i - row * kInElems;
            if (i < kInRows * kInElems) *reinterpret_cast<uint16_t*>(in_lds + row * kInStride + e * 2) = sv[it];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

        // ---- implicit GEMM, channels x pixels: wave w owns pixel tiles w and w + 4 (32 convolution outputs each) x both channel
        // tiles.  The WEIGHTS are the MFMA's row operand, so a lane ends up holding 2 x 16 channels of ONE pixel -----------------
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int rt = wave + 4 * half;
            const int m_raw = rt * 32 + nl;
            const int m = m_raw < kConvPix ? m_raw : kConvPix - 1;
            const int lr = m / kConvCols, lc = m - lr * kConvCols;
            const unsigned char* a_base = in_lds + (2 * lr) * kInStride + lc * 12;
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + (kg * 2 + ct) * 16 + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[ct][4 * q + i] = b4[i];
                }
#pragma unroll
            for (int s = 0; s < kKsteps; ++s) {
                // k-group G = 2 s + kg is tap row G / 3, elements 8 (G % 3) .. + 7 of its 24
                const int g0 = 2 * s, g1 = 2 * s + 1;
                const int off = kg ? (g1 / 3) * kInStride + (g1 % 3) * 16 : (g0 / 3) * kInStride + (g0 % 3) * 16;
                const unsigned* ap = reinterpret_cast<const unsigned*>(a_base + off);
                const u32x4 a = {ap[0], ap[1], ap[2], ap[3]};
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wreg[s][ct]), as_bf16x8(a), acc[ct], 0, 0, 0);
            }
            // ReLU -> bf16 into the convolution buffer, 4 consecutive channels per write; pixels outside the convolution's output
            // are the pool's padding (0 is neutral for a max over ReLU outputs)
            const int cy = cy0 + lr, cx = cx0 + lc;
            const bool ok = cy >= 0 && cy < dm.Hc && cx >= 0 && cx < dm.Wc;
            if (m_raw < kConvPix) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned lo = 0u, hi = 0u;
                        if (ok) {
                            const float v0 = acc[ct][4 * q], v1 = acc[ct][4 * q + 1], v2 = acc[ct][4 * q + 2], v3 = acc[ct][4 * q + 3];
                            lo = pack_bf16x2(v0 > 0.f ? v0 : 0.f, v1 > 0.f ? v1 : 0.f);
                            hi = pack_bf16x2(v2 > 0.f ? v2 : 0.f, v3 > 0.f ? v3 : 0.f);
                        }
                        *reinterpret_cast<u32x2*>(conv_lds + m * kConvStride + (32 * ct + 8 * q + 4 * kg) * 2) = u32x2{lo, hi};
                    }
            }
        }
        __syncthreads();

        // ---- 3x3 / stride 2 max-pool out of LDS: thread = (pooled pixel, 16 channels); non-negative bf16 order like uint16 -----
        if (tid < kPoolRows * kPoolCols * 4) {
            const int pp = tid >> 2, cg = tid & 3;
            const int lpy = pp / kPoolCols, lpx = pp - lpy * kPoolCols;
            u32x4 m0 = {0u, 0u, 0u, 0u}, m1 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const unsigned char* p = conv_lds + ((2 * lpy + dy) * kConvCols + 2 * lpx + dx) * kConvStride + cg * 32;
                    const u32x4 v0 = *reinterpret_cast<const u32x4*>(p), v1 = *reinterpret_cast<const u32x4*>(p + 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        m0[j] = max_u16x2(m0[j], v0[j]);
                        m1[j] = max_u16x2(m1[j], v1[j]);
                    }
                }
            const int py = py0 + lpy, px = px0 + lpx;
            if (py < dm.Hp && px < dm.Wp) {
                bf16_t* o = Y + (((size_t)n * dm.Hp + py) * dm.Wp + px) * 64 + cg * 16;
                *reinterpret_cast<u32x4*>(o) = m0;
                *reinterpret_cast<u32x4*>(o + 8) = m1;
            }
        }
        __syncthreads();  // both buffers are rewritten by the next tile
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" int alo_stem_conv_pool(const void* x, const void* w_packed, const void* bias, void* y, int N, int H, int W,
                                  long stride_n, long stride_c, long stride_h, long stride_w, int dtype, void* stream) {
    ALO_REQUIRE(x && w_packed && y, ALO_ERR_INVALID_ARGUMENT, "alo_stem_conv_pool: null pointer argument");
    ALO_REQUIRE(N > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_stem_conv_pool: N, H, W must be positive");
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_stem_conv_pool: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)w_packed | (uintptr_t)y) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_stem_conv_pool: w_packed and y must be 16-byte aligned");
    StemDims dm;
    dm.N = N; dm.H = H; dm.W = W;
    dm.Hc = (H - 1) / 2 + 1; dm.Wc = (W - 1) / 2 + 1;        // (H + 6 - 7) / 2 + 1
    dm.Hp = (dm.Hc - 1) / 2 + 1; dm.Wp = (dm.Wc - 1) / 2 + 1;  // (Hc + 2 - 3) / 2 + 1
    dm.sN = stride_n; dm.sC = stride_c; dm.sH = stride_h; dm.sW = stride_w;
    dm.tiles_y = (dm.Hp + kPoolRows - 1) / kPoolRows;
    dm.tiles_x = (dm.Wp + kPoolCols - 1) / kPoolCols;
    const long ntiles = (long)dm.tiles_y * dm.tiles_x * N;
    ALO_REQUIRE(ntiles < (1L << 30), ALO_ERR_UNSUPPORTED, "alo_stem_conv_pool: image batch too large");
    const void* kern = reinterpret_cast<const void*>(stem_conv_pool_kernel);
    void* args[] = {&x, &w_packed, &bias, &y, &dm};
    const unsigned grid = (unsigned)(ntiles < 768 ? ntiles : 768);   // 256 CUs x 3 resident workgroups
    hipError_t e = hipLaunchKernel(kern, dim3(grid), dim3(kStemThreads), args, kStemLds, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_stem_conv_pool: %s", hipGetErrorString(e));
    return check_launch("alo_stem_conv_pool");
}
