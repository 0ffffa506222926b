// 3x3 convolution, stride 1, padding 1, NHWC bf16 (fp32 accumulation) with bias + ReLU in the epilogue: the middle
// convolution of the ResNet bottlenecks (alonet/detr/backbone.py + torchvision Bottleneck.conv2, FrozenBatchNorm folded in).
//
// An implicit GEMM on v_mfma_f32_32x32x16_bf16: y[p, :] = sum over the 9 taps of x[p + tap shift, :] W_tap^T, K = 9 * Cin.
//   * a workgroup owns 64 consecutive pixels of one image and 128 output channels (wave w: channels 32 w ..);
//   * the input it needs — three runs of 66 pixels (rows y-1, y, y+1 around the tile) — is copied once per 128-channel chunk
//     into LDS with fully coalesced loads; A fragments are ds_read from there with the tap's pixel shift, and zeroed per
//     lane where the tap falls outside the image (the padding);
//   * the weights arrive pre-packed in MFMA B-fragment order ([Cout / 32][9 Cin / 16][64 lanes][8], k = tap * Cin + c) and
//     stream through registers two k-steps ahead, as in ffn256_kernel.
#include "common.hpp"

namespace alo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

constexpr int kPix = 64;            // pixels per tile
constexpr int kChunk = 64;          // input channels staged at once
constexpr int kBpt = kChunk / 32;   // weight batches (of two k-steps) per tap and chunk
constexpr int kSegPix = kPix + 2;   // pixels per halo run
constexpr int kPixStride = kChunk * 2 + 16;  // LDS bytes per staged pixel (+16: conflict-free 16-byte fragment reads)
constexpr int kOutStride = 64 * 2 + 16;      // LDS bytes per pixel of a wave's 64 x 64 output block

struct ConvDims {
    int N, H, W, Cin, Cout;
    int tiles_per_image;
};

constexpr int kThreads = 128;       // two waves: wave w owns output channels 64 w .. 64 w + 63 of the block's 128

struct WFrag { u32x4 v[2][2]; };    // [k-step of the batch][column tile]

template <bool RELU>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ bias,
               bf16_t* __restrict__ Y, const ConvDims dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;                                   // [3 runs][66 pixels][kPixStride]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char* const obuf = smem + wave * (kPix * kOutStride);      // aliases the halo once the K loop is over
    const int nl = lane & 31, kg = lane >> 5;
    const int HW = dm.H * dm.W;
    const int img = blockIdx.x / dm.tiles_per_image;
    const int p0 = (blockIdx.x % dm.tiles_per_image) * kPix;           // first pixel of the tile inside the image
    const int col0 = blockIdx.y * 128 + wave * 64;                      // this wave's 64 output channels (two column tiles)
    const bool has_cols = col0 < dm.Cout;                               // Cout % 64 == 0
    const bf16_t* ximg = X + (size_t)img * HW * dm.Cin;

    // which of the 9 taps exist for this lane's two pixels (row tiles a = 0, 1: pixel p0 + 32 a + nl)
    unsigned tapmask[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int q = p0 + 32 * a + nl;
        const int y = q / dm.W, x = q - y * dm.W;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (q < HW && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) m |= 1u << t;
        }
        tapmask[a] = m;
    }

    f32x16 acc[2][2];  // [row tile][column tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int ksteps_per_tap = dm.Cin / 16;
    const size_t ctile_stride = (size_t)9 * ksteps_per_tap * 512;      // elements between the packed column tiles
    const bf16_t* wfrag = Wp + (size_t)(col0 / 32) * ctile_stride + lane * 8;

    // the halo of one chunk: 3 runs x 66 pixels x 64 channels = 1584 pieces of 16 B; pixels outside the image are clamped (their
    // taps are masked at use).  Loaded into registers one chunk AHEAD, so the round trip hides behind the previous chunk's MFMAs.
    constexpr int kPieces = 3 * kSegPix * (kChunk / 8), kIters = (kPieces + kThreads - 1) / kThreads;
    u32x4 stage[kIters];
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int i = tid + it * kThreads;
            const int piece = i % (kChunk / 8), pr = i / (kChunk / 8), run = pr / kSegPix, pix = pr - run * kSegPix;
            int q = p0 + (run - 1) * dm.W - 1 + pix;
            q = q < 0 ? 0 : (q >= HW ? HW - 1 : q);
            if (i < kPieces) stage[it] = *reinterpret_cast<const u32x4*>(ximg + (size_t)q * dm.Cin + c0 + piece * 8);
        }
    };
    load_halo(0);

    for (int c0 = 0; c0 < dm.Cin; c0 += kChunk) {
        // weights of batch kb (two k-steps) of this chunk: tap kb / kBpt, k-steps 2 (kb % kBpt), + 1 of the chunk
        auto load_w = [&](WFrag& f, int kb) {
            const bf16_t* p = wfrag + (size_t)((kb / kBpt) * ksteps_per_tap + c0 / 16 + 2 * (kb % kBpt)) * 512;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    f.v[j][b] = *reinterpret_cast<const u32x4*>(p + (size_t)j * 512 + b * ctile_stride);
        };
        auto compute = [&](const WFrag& f, int kb) {
            const int t = kb / kBpt, run = t / 3, dx = t - 3 * run;
            const bool ok0 = (tapmask[0] >> t) & 1u, ok1 = (tapmask[1] >> t) & 1u;
            const unsigned char* a_base = halo + (run * kSegPix + dx + nl) * kPixStride + kg * 16 + (kb % kBpt) * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 a0 = *reinterpret_cast<const u32x4*>(a_base + j * 32);
                u32x4 a1 = *reinterpret_cast<const u32x4*>(a_base + 32 * kPixStride + j * 32);
                if (!ok0) a0 = u32x4{0u, 0u, 0u, 0u};
                if (!ok1) a1 = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(f.v[j][b]), acc[0][b], 0, 0, 0);
                    acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(f.v[j][b]), acc[1][b], 0, 0, 0);
                }
            }
        };
        WFrag w0, w1;
        if (has_cols) load_w(w0, 0);

#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int i = tid + it * kThreads;
            const int piece = i % (kChunk / 8), pr = i / (kChunk / 8);
            if (i < kPieces) *reinterpret_cast<u32x4*>(halo + pr * kPixStride + piece * 16) = stage[it];
        }
        __syncthreads();
        if (c0 + kChunk < dm.Cin) load_halo(c0 + kChunk);

        if (has_cols) {
            constexpr int kBatches = 9 * kBpt;
#pragma unroll 1
            for (int kb = 0; kb < kBatches; kb += 2) {
                load_w(w1, kb + 1);
                compute(w0, kb);
                if (kb + 2 < kBatches) load_w(w0, kb + 2);
                compute(w1, kb + 1);
            }
        }
        __syncthreads();  // the halo is overwritten by the next channel chunk
    }

    if (has_cols) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float bv = bias ? bf16_to_f32(bias[col0 + 32 * b + nl].bits) : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float v = acc[a][b][r] + bv;
                    if (RELU) v = fmaxf(v, 0.f);
                    *reinterpret_cast<uint16_t*>(obuf + row * kOutStride + (32 * b + nl) * 2) = f32_to_bf16(v);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 64 pixels x 128 B: 8 lanes x 16 B per pixel, 8 pixels per store instruction
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 8 + (lane >> 3);
            const int q = p0 + row;
            if (q < HW)
                *reinterpret_cast<u32x4*>(Y + ((size_t)img * HW + q) * dm.Cout + col0 + (lane & 7) * 8) =
                    *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + (lane & 7) * 16);
        }
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" int alo_conv3x3_nhwc(const void* x, const void* w_packed, const void* bias, void* y, int N, int H, int W, int Cin,
                                int Cout, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && w_packed && y, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_nhwc: null pointer argument");
    ALO_REQUIRE(N > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_nhwc: N, H, W must be positive");
    ALO_REQUIRE(Cin >= kChunk && Cin % kChunk == 0 && Cout >= 64 && Cout % 64 == 0, ALO_ERR_UNSUPPORTED,
                "alo_conv3x3_nhwc: Cin must be a multiple of %d and Cout a multiple of 64 (Cin=%d Cout=%d)", kChunk, Cin, Cout);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_conv3x3_nhwc: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_conv3x3_nhwc: pointers must be 16-byte aligned");
    ConvDims dm;
    dm.N = N; dm.H = H; dm.W = W; dm.Cin = Cin; dm.Cout = Cout;
    dm.tiles_per_image = (H * W + kPix - 1) / kPix;
    const size_t lds = 3 * kSegPix * kPixStride > 2 * kPix * kOutStride ? 3 * kSegPix * kPixStride : 2 * kPix * kOutStride;  // halo, aliased by the output staging
    void* args[] = {&x, &w_packed, &bias, &y, &dm};
    const void* kern = relu ? reinterpret_cast<const void*>(conv3x3_kernel<true>) : reinterpret_cast<const void*>(conv3x3_kernel<false>);
    static bool attr_set[2] = {false, false};
    if (!attr_set[relu ? 1 : 0]) {
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[relu ? 1 : 0] = true;
    }
    hipError_t e = hipLaunchKernel(kern, dim3((unsigned)(dm.tiles_per_image * N), (unsigned)((Cout + 127) / 128)), dim3(kThreads), args,
                                   lds, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_conv3x3_nhwc: %s", hipGetErrorString(e));
    return check_launch("alo_conv3x3_nhwc");
}
