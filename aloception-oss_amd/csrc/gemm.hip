// Y (M, N) = act(X (M, K) @ W (N, K)^T + bias), K in {64, 128, 256}, bf16 with fp32 accumulation: the short-K linear layers
// around the attention op — MSDeformAttn's value_proj / sampling_offsets / attention_weights / output_proj and the FFN's
// first layer over the encoder's 177784 rows, the backbone's 1x1 convolutions with few input channels over NHWC rows.
//
// These products are memory bound (23 GFLOP against 182 MB at N = 256) and the library kernel PyTorch reaches streams them
// at 2.8 TB/s.  This kernel is built around the stream instead of around the tile:
//   * the WEIGHTS never move: each of a workgroup's 4 waves keeps its 64 output columns of W (64 x 256 bf16 = 128 VGPRs) in
//     registers for the whole launch, already in the lane order v_mfma_f32_32x32x16_bf16 wants for its B operand;
//   * X streams through: persistent workgroups walk 32-row tiles; a tile is fetched with fully coalesced 16-byte loads
//     (one 512-byte row per 32 lanes), parked in LDS (double buffered, next tile in flight during the MFMAs) and read back
//     as A fragments by all four waves;
//   * Y leaves through LDS as well, so that every store instruction writes whole 128-byte lines (the MFMA accumulator
//     layout has a lane own one column and 16 scattered rows).
#include "common.hpp"

namespace alo {
namespace {

constexpr int kRows = 32;               // rows of X per tile
constexpr int kOutStride = 64 * 2 + 16; // LDS row stride of a wave's 32 x 64 output block

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

struct GemmDims {
    long M;
    int N;
    int tiles;  // ceil(M / 32)
};

template <int kK, bool RELU>
__global__ void __launch_bounds__(256)
linear_shortk_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
                   bf16_t* __restrict__ Y, const GemmDims dm) {
    constexpr int kRowBytes = kK * 2 + 16;  // LDS row stride of the X tile: +16 B keeps the 16-byte fragment reads conflict-free
    constexpr int kSteps = kK / 16;         // MFMA k-steps per tile
    constexpr int kPieces = kK / 8;         // 16-byte pieces per row
    constexpr int kLoads = kRows * kPieces / 256 > 0 ? kRows * kPieces / 256 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xbuf0 = smem;
    unsigned char* const xbuf1 = smem + kRows * kRowBytes;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char* obuf = smem + 2 * kRows * kRowBytes + wave * (kRows * kOutStride);  // per-wave output staging

    const int col0 = blockIdx.y * 256 + wave * 64;  // this wave's 64 output columns
    const bool has_cols = col0 < dm.N;              // N is a multiple of 64
    const int nl = lane & 31, kg = lane >> 5;       // MFMA lane roles: row/column inside a 32-tile, 8-element k group

    // ---- resident B operand: W[col0 + 32 t + nl][16 s + 8 kg .. + 8) for t < 2, s < 16 ------------------------------------
    u32x4 wreg[2][kSteps];
    float bias_v[2] = {0.f, 0.f};
    if (has_cols) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16_t* wr = W + (size_t)(col0 + 32 * t + nl) * kK + 8 * kg;
#pragma unroll
            for (int s = 0; s < kSteps; ++s) wreg[t][s] = *reinterpret_cast<const u32x4*>(wr + 16 * s);
            if (bias != nullptr) bias_v[t] = bf16_to_f32(bias[col0 + 32 * t + nl].bits);
        }
    }

    // ---- tile loader: 32 rows x 2K bytes in 16-byte pieces; thread -> (row, piece) keeps rows contiguous -------------------
    auto fetch = [&](int tile, u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;  // piece index: row = p / kPieces, 16-byte column = p % kPieces
            const long row = (long)tile * kRows + p / kPieces;
            r[j] = (p < kRows * kPieces && row < dm.M) ? *reinterpret_cast<const u32x4*>(X + row * kK + (p % kPieces) * 8)
                                                        : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto park = [&](unsigned char* buf, const u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;
            if (p < kRows * kPieces) *reinterpret_cast<u32x4*>(buf + (p / kPieces) * kRowBytes + (p % kPieces) * 16) = r[j];
        }
    };

    int tile = blockIdx.x;
    if (tile >= dm.tiles) return;
    u32x4 stage[kLoads];
    fetch(tile, stage);
    park(xbuf0, stage);
    __syncthreads();

    for (int it = 0;; ++it) {
        const int next = tile + gridDim.x;
        const bool more = next < dm.tiles;
        if (more) fetch(next, stage);  // in flight during the MFMAs below
        const unsigned char* xb = (it & 1) ? xbuf1 : xbuf0;

        if (has_cols) {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                // A fragment: X[tile row nl][16 s + 8 kg .. + 8)
                const u32x4 a = *reinterpret_cast<const u32x4*>(xb + nl * kRowBytes + (16 * s + 8 * kg) * 2);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(wreg[0][s]), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(wreg[1][s]), acc[1], 0, 0, 0);
            }
            // accumulator (lane = column nl, register r = row (r & 3) + 8 (r >> 2) + 4 kg) -> + bias -> bf16 -> LDS [row][col]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float v = acc[t][r] + bias_v[t];
                    if (RELU) v = fmaxf(v, 0.f);
                    *reinterpret_cast<uint16_t*>(obuf + row * kOutStride + (32 * t + nl) * 2) = f32_to_bf16(v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // rows leave as whole 128-byte lines: 8 lanes x 16 B per row, 8 rows per store instruction
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + (lane >> 3);
                const long grow = (long)tile * kRows + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + (lane & 7) * 16);
                if (grow < dm.M) *reinterpret_cast<u32x4*>(Y + grow * dm.N + col0 + (lane & 7) * 8) = v;
            }
        }
        if (!more) break;
        park((it & 1) ? xbuf0 : xbuf1, stage);  // the buffer read two iterations ago: every wave passed the barrier below since
        __syncthreads();
        tile = next;
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

namespace {
template <int K, bool RELU>
int launch_shortk(const void* x, const void* weight, const void* bias, void* y, long M, int N, hipStream_t stream) {
    GemmDims dm;
    dm.M = M; dm.N = N; dm.tiles = (int)((M + kRows - 1) / kRows);
    const size_t lds = 2 * kRows * (K * 2 + 16) + 4 * kRows * kOutStride;
    const int cols = (N + 255) / 256;
    int gx = 512 / cols;  // persistent: about two workgroups per CU in total
    if (gx > dm.tiles) gx = dm.tiles;
    if (gx < 1) gx = 1;
    void* args[] = {&x, &weight, &bias, &y, &dm};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(linear_shortk_kernel<K, RELU>), dim3(gx, cols), dim3(256), args,
                                   lds, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_linear_shortk: %s", hipGetErrorString(e));
    return check_launch("alo_linear_shortk");
}
}  // namespace

extern "C" int alo_linear_shortk(const void* x, const void* weight, const void* bias, void* y, long M, int N, int K, int relu,
                                 int dtype, void* stream) {
    ALO_REQUIRE(x && weight && y, ALO_ERR_INVALID_ARGUMENT, "alo_linear_shortk: null pointer argument");
    ALO_REQUIRE(M > 0 && N > 0 && N % 64 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_linear_shortk: M must be positive and N a positive multiple of 64 (M=%ld N=%d)", M, N);
    ALO_REQUIRE(K == 64 || K == 128 || K == 256, ALO_ERR_UNSUPPORTED, "alo_linear_shortk: K must be 64, 128 or 256, got %d", K);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_linear_shortk: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_linear_shortk: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define ALO_GEMM_CASE(KK)                                                                       \
    if (K == KK) return relu ? launch_shortk<KK, true>(x, weight, bias, y, M, N, st)           \
                             : launch_shortk<KK, false>(x, weight, bias, y, M, N, st);
    ALO_GEMM_CASE(64) ALO_GEMM_CASE(128) ALO_GEMM_CASE(256)
#undef ALO_GEMM_CASE
    return ALO_ERR_UNSUPPORTED;
}
