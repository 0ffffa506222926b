// Y (M, N) = act(X (M, K) @ W (N, K)^T + bias [+ R]), bf16 with fp32 accumulation, K a multiple of 256, N of 128: the backbone's 1x1
// convolutions with many input channels (512 / 1024 / 2048; alonet/detr/backbone.py:19-47 + torchvision Bottleneck.conv1 / conv3 /
// downsample) and the 1x1 input projections (alonet/deformable_detr/deformable_detr.py:75-84) over NHWC rows.
//
// linear_shortk_kernel keeps W in registers; at these K it cannot.  Here W streams instead — pre-packed in MFMA fragment order
// (alo_pack_mfma_b: one 1 KB line per fragment), two k-steps ahead of its use through two register buffers, as in ffn256_kernel —
// while X goes through LDS 256 (128) columns at a time, the next chunk being fetched into registers during the MFMAs of the
// current one.  A wave owns a 64 x 64 block of Y (2 x 2 MFMA tiles, computed transposed so a lane holds one row); a workgroup
// is 1 x 4 waves (64 rows x 256 columns) or, for N = 128, 2 x 2 (128 rows x 128 columns).
#include "common.hpp"

namespace alo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

constexpr int kOutStride = 64 * 2 + 16;   // LDS row stride of a wave's 64 x 64 output block

struct PackedDims {
    long M;
    int N, K;
    // strided 1x1 convolution: row r of X' = pixel (n, gs * yo, gs * xo) of the NHWC map X; gs <= 1: X' = X
    int gs, gWo, gHoWo, gW, gHW;
};

__device__ __forceinline__ long gather_row(const PackedDims& dm, long row) {
    if (dm.gs <= 1) return row;
    const long n = row / dm.gHoWo;
    const int rem = (int)(row - n * dm.gHoWo);
    const int yo = rem / dm.gWo, xo = rem - yo * dm.gWo;
    return n * dm.gHW + (long)(yo * dm.gs) * dm.gW + xo * dm.gs;
}

template <int WC, bool RELU, bool HAS_RES>
__global__ void __launch_bounds__(256, 2)
linear_packed_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ bias,
                     const bf16_t* __restrict__ R, bf16_t* __restrict__ Y, const PackedDims dm) {
    constexpr int WR = 4 / WC;                 // waves along the rows
    constexpr int kRows = 64 * WR;             // rows of X per workgroup
    constexpr int KC = 256 / WR;               // columns of X staged at once
    constexpr int kStride = KC * 2 + 16;       // LDS row stride of the X chunk (+16 B: conflict-free fragment reads)
    constexpr int kPieces = KC / 8;            // 16-byte pieces per staged row
    constexpr int kLoads = kRows * kPieces / 256;   // = 8
    constexpr int KB = 2;                      // k-steps per weight batch
    constexpr int NB = KC / 16 / KB;           // batches per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xs = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nl = lane & 31, kg = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const long row0 = (long)blockIdx.x * kRows;
    const int col0 = blockIdx.y * (64 * WC) + wc * 64;   // this wave's 64 output columns (N % (64 WC) == 0)
    constexpr int kXBytes = kRows * kStride, kOBytes = 4 * 64 * kOutStride;
    float* const bias_s = reinterpret_cast<float*>(smem + (kXBytes > kOBytes ? kXBytes : kOBytes));   // [WC][64], behind both uses of the buffer
    unsigned char* const obuf = smem + wave * (64 * kOutStride);              // aliases the X chunk after the K loop

    bias_s[tid] = bias != nullptr ? bf16_to_f32(bias[blockIdx.y * (64 * WC) + (tid & (64 * WC - 1))].bits) : 0.f;

    auto fetch = [&](int chunk, u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;
            long row = row0 + p / kPieces;
            row = gather_row(dm, row < dm.M ? row : dm.M - 1);   // rows past the end are read from the last row and never stored
            r[j] = *reinterpret_cast<const u32x4*>(X + row * dm.K + chunk * KC + (p % kPieces) * 8);
        }
    };
    auto park = [&](const u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;
            *reinterpret_cast<u32x4*>(xs + (p / kPieces) * kStride + (p % kPieces) * 16) = r[j];
        }
    };
    const int ksteps = dm.K / 16;
    const size_t tile_stride = (size_t)ksteps * 512;   // elements between packed column tiles
    const bf16_t* wfrag = Wp + (size_t)(col0 / 32) * tile_stride + lane * 8;
    auto load_batch = [&](u32x4 (&buf)[2][KB], int gbatch) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < KB; ++j)
                buf[t][j] = *reinterpret_cast<const u32x4*>(wfrag + t * tile_stride + (size_t)(KB * gbatch + j) * 512);
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc[2][2];   // [row tile][column tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][t][i] = 0.f;
    const unsigned char* a_lds = xs + (64 * wr) * kStride;
    auto mma_batch = [&](const u32x4 (&buf)[2][KB], int batch) {
        u32x4 af[2][KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int s = KB * batch + j;
            af[0][j] = *reinterpret_cast<const u32x4*>(a_lds + nl * kStride + (16 * s + 8 * kg) * 2);
            af[1][j] = *reinterpret_cast<const u32x4*>(a_lds + (32 + nl) * kStride + (16 * s + 8 * kg) * 2);
        }
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(buf[t][j]), as_bf16x8(af[a][j]), acc[a][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nchunks = dm.K / KC;
    const int total_batches = nchunks * NB;
    u32x4 stage[kLoads];
    u32x4 bufa[2][KB], bufb[2][KB];
    fetch(0, stage);
    load_batch(bufa, 0);
    for (int c = 0; c < nchunks; ++c) {
        park(stage);
        __syncthreads();
        if (c + 1 < nchunks) fetch(c + 1, stage);   // in flight during this chunk's MFMAs
        const int gb = c * NB;
#pragma unroll
        for (int bt = 0; bt < NB; bt += 2) {
            load_batch(bufb, gb + bt + 1);
            mma_batch(bufa, bt);
            if (gb + bt + 2 < total_batches) load_batch(bufa, gb + bt + 2);
            mma_batch(bufb, bt + 1);
        }
        __syncthreads();   // everyone is done with the chunk (and, after the last one, with the LDS it sits in)
    }

    // ---- epilogue: lane = row 64 wr + 32 a + nl; registers 4 q .. 4 q + 3 = columns col0 + 32 t + 8 q + 4 kg .. + 3 ----------------
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 32 * t + 8 * q + 4 * kg;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + wc * 64 + c);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float v0 = acc[a][t][4 * q] + bb[0], v1 = acc[a][t][4 * q + 1] + bb[1];
                float v2 = acc[a][t][4 * q + 2] + bb[2], v3 = acc[a][t][4 * q + 3] + bb[3];
                if (RELU && !HAS_RES) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                *reinterpret_cast<u32x2*>(obuf + (32 * a + nl) * kOutStride + c * 2) = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // rows leave as whole 128-byte lines: 8 lanes x 16 B per row, 8 rows per store instruction
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 8 + (lane >> 3);
        const long grow = row0 + 64 * wr + row;
        u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + (lane & 7) * 16);
        if (grow < dm.M) {
            if constexpr (HAS_RES) {   // + identity (same coordinates as y), then the activation
                const u32x4 rv = *reinterpret_cast<const u32x4*>(R + grow * dm.N + col0 + (lane & 7) * 8);
                const unsigned a4[4] = {v.x, v.y, v.z, v.w}, r4[4] = {rv.x, rv.y, rv.z, rv.w};
                unsigned o4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = __uint_as_float(a4[i] << 16) + __uint_as_float(r4[i] << 16);
                    float hi = __uint_as_float(a4[i] & 0xffff0000u) + __uint_as_float(r4[i] & 0xffff0000u);
                    if (RELU) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                    o4[i] = pack_bf16x2(lo, hi);
                }
                v = u32x4{o4[0], o4[1], o4[2], o4[3]};
            }
            *reinterpret_cast<u32x4*>(Y + grow * dm.N + col0 + (lane & 7) * 8) = v;
        }
    }
}

template <int WC, bool RELU, bool HAS_RES>
int launch_packed(const void* x, const void* w, const void* bias, const void* residual, void* y, const PackedDims& dm, hipStream_t stream) {
    constexpr int WR = 4 / WC, kRows = 64 * WR, KC = 256 / WR;
    constexpr size_t xbytes = (size_t)kRows * (KC * 2 + 16), obytes = 4 * 64 * kOutStride;
    constexpr size_t lds = (xbytes > obytes ? xbytes : obytes) + 256 * sizeof(float);
    static_assert(xbytes >= obytes || true, "");
    const void* kern = reinterpret_cast<const void*>(linear_packed_kernel<WC, RELU, HAS_RES>);
    void* args[] = {&x, &w, &bias, &residual, &y, const_cast<PackedDims*>(&dm)};
    hipError_t e = hipLaunchKernel(kern, dim3((unsigned)((dm.M + kRows - 1) / kRows), (unsigned)(dm.N / (64 * WC))), dim3(256), args,
                                   lds, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_linear_packed: %s", hipGetErrorString(e));
    return check_launch("alo_linear_packed");
}

}  // namespace
}  // namespace alo

using namespace alo;

namespace {
int dispatch_packed(const void* x, const void* w_packed, const void* bias, const void* residual, void* y, const PackedDims& dm,
                    int relu, void* stream);
}

extern "C" int alo_linear_packed(const void* x, const void* w_packed, const void* bias, const void* residual, void* y, long M,
                                 int N, int K, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && w_packed && y, ALO_ERR_INVALID_ARGUMENT, "alo_linear_packed: null pointer argument");
    ALO_REQUIRE(M > 0 && N > 0 && K > 0, ALO_ERR_INVALID_ARGUMENT, "alo_linear_packed: sizes must be positive");
    ALO_REQUIRE(K % 256 == 0 && N % 128 == 0, ALO_ERR_UNSUPPORTED,
                "alo_linear_packed: K must be a multiple of 256 and N of 128 (K=%d N=%d)", K, N);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_linear_packed: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_linear_packed: pointers must be 16-byte aligned");
    ALO_REQUIRE((M + 63) / 64 < (1L << 31), ALO_ERR_UNSUPPORTED, "alo_linear_packed: too many rows");
    PackedDims dm;
    dm.M = M; dm.N = N; dm.K = K;
    dm.gs = 1; dm.gWo = dm.gHoWo = dm.gW = dm.gHW = 1;
    return dispatch_packed(x, w_packed, bias, residual, y, dm, relu, stream);
}

extern "C" int alo_internal_shortk_gather(const void* x, const void* weight, const void* bias, const void* residual, void* y,
                                          long M, int N, int K, int relu, const int* gather, void* stream);

extern "C" int alo_conv1x1_nhwc(const void* x, const void* weight, int weight_is_packed, const void* bias, const void* residual,
                                void* y, int N, int H, int W, int Cin, int Cout, int stride, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && weight && y, ALO_ERR_INVALID_ARGUMENT, "alo_conv1x1_nhwc: null pointer argument");
    ALO_REQUIRE(N > 0 && H > 0 && W > 0 && stride >= 1, ALO_ERR_INVALID_ARGUMENT, "alo_conv1x1_nhwc: N, H, W, stride must be positive");
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_conv1x1_nhwc: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_conv1x1_nhwc: pointers must be 16-byte aligned");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long M = (long)N * Ho * Wo;
    ALO_REQUIRE((long)H * W < (1L << 30), ALO_ERR_UNSUPPORTED, "alo_conv1x1_nhwc: map too large");
    const int gather[5] = {stride, Ho, Wo, H, W};
    if (!weight_is_packed) {
        ALO_REQUIRE(Cout % 64 == 0, ALO_ERR_UNSUPPORTED, "alo_conv1x1_nhwc: Cout must be a multiple of 64 (got %d)", Cout);
        return alo_internal_shortk_gather(x, weight, bias, residual, y, M, Cout, Cin, relu, gather, stream);
    }
    ALO_REQUIRE(Cin % 256 == 0 && Cout % 128 == 0, ALO_ERR_UNSUPPORTED,
                "alo_conv1x1_nhwc: packed weights need Cin %% 256 == 0 and Cout %% 128 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    PackedDims dm;
    dm.M = M; dm.N = Cout; dm.K = Cin;
    dm.gs = stride; dm.gWo = Wo; dm.gHoWo = Ho * Wo; dm.gW = W; dm.gHW = H * W;
    return dispatch_packed(x, weight, bias, residual, y, dm, relu, stream);
}

namespace {
int dispatch_packed(const void* x, const void* w_packed, const void* bias, const void* residual, void* y, const PackedDims& dm,
                    int relu, void* stream) {
    const int N = dm.N;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool r = relu != 0, res = residual != nullptr;
    if (N % 256 == 0) {
        if (res) return r ? launch_packed<4, true, true>(x, w_packed, bias, residual, y, dm, s) : launch_packed<4, false, true>(x, w_packed, bias, residual, y, dm, s);
        return r ? launch_packed<4, true, false>(x, w_packed, bias, residual, y, dm, s) : launch_packed<4, false, false>(x, w_packed, bias, residual, y, dm, s);
    }
    if (res) return r ? launch_packed<2, true, true>(x, w_packed, bias, residual, y, dm, s) : launch_packed<2, false, true>(x, w_packed, bias, residual, y, dm, s);
    return r ? launch_packed<2, true, false>(x, w_packed, bias, residual, y, dm, s) : launch_packed<2, false, false>(x, w_packed, bias, residual, y, dm, s);
}
}  // namespace
