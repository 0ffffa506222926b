// Y (M, N) = act(X (M, K) @ W (N, K)^T + bias), K in {64, 128, 256}, bf16 with fp32 accumulation: the short-K linear layers
// around the attention op — MSDeformAttn's value_proj / sampling_offsets / attention_weights / output_proj and the FFN's
// first layer over the encoder's 177784 rows, the backbone's 1x1 convolutions with few input channels over NHWC rows.
//
// These products are memory bound (23 GFLOP against 182 MB at N = 256) and the library kernel PyTorch reaches streams them
// at 2.8 TB/s.  This kernel is built around the stream instead of around the tile:
//   * the WEIGHTS never move: each of a workgroup's 4 waves keeps its 64 output columns of W (64 x 256 bf16 = 128 VGPRs) in
//     registers for the whole launch, already in the lane order v_mfma_f32_32x32x16_bf16 wants for its B operand;
//   * X streams through: persistent workgroups walk 64-row tiles; a tile is fetched with fully coalesced 16-byte loads
//     (one 512-byte row per 32 lanes) into registers while the previous tile is being multiplied, parked in LDS and read
//     back as A fragments by all four waves;
//   * Y leaves through LDS as well, so that every store instruction writes whole 128-byte lines (the MFMA accumulator
//     layout has a lane own one column and 16 scattered rows).
#include "common.hpp"

namespace alo {
namespace {

constexpr int kRows = 64;               // rows of X per tile
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
    int tiles;  // ceil(M / kRows)
    int S;      // head-major output only: rows per batch item
    // strided 1x1 convolution: row r of X' = pixel (n, gs * yo, gs * xo) of the NHWC map X; gs <= 1: X' = X
    int gs, gWo, gHoWo, gW, gHW;
};

// row of the GEMM -> row of the NHWC input it reads (identity unless the 1x1 convolution is strided)
template <typename D>
__device__ __forceinline__ long gather_row(const D& dm, long row) {
    if (dm.gs <= 1) return row;
    const long n = row / dm.gHoWo;
    const int rem = (int)(row - n * dm.gHoWo);
    const int yo = rem / dm.gWo, xo = rem - yo * dm.gWo;
    return n * dm.gHW + (long)(yo * dm.gs) * dm.gW + xo * dm.gs;
}

// HM = true (value_proj of MSDeformAttn): y is written HEAD-major, (batch, N / 32 heads, S, 32), and rows whose padding-mask
// byte is set are written as zeros — `value.masked_fill(mask, 0)` and the re-layout the head-major attention kernel wants,
// for free in the epilogue (a wave's 64 columns are two heads; 16 rows of a head are 1 KB contiguous).  R then carries the
// (M,) uint8 mask (or NULL) and dm.S the rows per batch item.
template <int kK, bool RELU, bool HAS_RES, bool HM>
__global__ void __launch_bounds__(256, 2)
linear_shortk_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias,
                     const bf16_t* __restrict__ R, bf16_t* __restrict__ Y, const GemmDims dm) {
    constexpr int kRowBytes = kK * 2 + 16;  // LDS row stride of the X tile: +16 B keeps the 16-byte fragment reads conflict-free
    constexpr int kSteps = kK / 16;         // MFMA k-steps per tile
    constexpr int kPieces = kK / 8;         // 16-byte pieces per row
    constexpr int kLoads = kRows * kPieces / 256;
    static_assert(kRows * kPieces % 256 == 0, "the tile loader gives every thread the same number of pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xbuf = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char* obuf = smem + kRows * kRowBytes + wave * (kRows * kOutStride);  // per-wave output staging

    const int col0 = blockIdx.y * 256 + wave * 64;  // this wave's 64 output columns
    const bool has_cols = col0 < dm.N;              // N is a multiple of 64
    const int nl = lane & 31, kg = lane >> 5;       // MFMA lane roles: row/column inside a 32-tile, 8-element k group

    // ---- resident B operand: W[col0 + 32 t + nl][16 s + 8 kg .. + 8) for t < 2, s < kSteps ----------------------------------
    // W is the MFMA's ROW operand (the product is computed transposed, Y^T = W X^T), so that a lane ends up holding 2 x 16
    // output columns of ONE row of the tile, four consecutive columns per accumulator quad: they leave as 8-byte LDS writes.
    u32x4 wreg[2][kSteps];
    // bias in accumulator order, [kg][t][16]: register r of lane (nl, kg) is column 32 t + (r & 3) + 8 (r >> 2) + 4 kg; the
    // accumulators start from it
    float* const bias_tab = reinterpret_cast<float*>(smem + kRows * kRowBytes + 4 * (kRows * kOutStride)) + wave * 64;
    if (has_cols) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16_t* wr = W + (size_t)(col0 + 32 * t + nl) * kK + 8 * kg;
#pragma unroll
            for (int s = 0; s < kSteps; ++s) wreg[t][s] = *reinterpret_cast<const u32x4*>(wr + 16 * s);
        }
        const int r = lane & 15, t = (lane >> 4) & 1, k2 = lane >> 5;
        bias_tab[lane] = bias != nullptr ? bf16_to_f32(bias[col0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * k2].bits) : 0.f;
    }

    // ---- tile loader: kRows rows x 2K bytes in 16-byte pieces; thread -> (row, piece) keeps rows contiguous.  Rows past the
    // end are read from the last row (and never stored), so the requests of a tile go out back to back, unpredicated -----------
    auto fetch = [&](int tile, u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;  // piece index: row = p / kPieces, 16-byte column = p % kPieces
            long row = (long)tile * kRows + p / kPieces;
            row = gather_row(dm, row < dm.M ? row : dm.M - 1);
            r[j] = *reinterpret_cast<const u32x4*>(X + row * kK + (p % kPieces) * 8);
        }
    };
    auto park = [&](const u32x4 (&r)[kLoads]) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int p = tid + 256 * j;
            *reinterpret_cast<u32x4*>(xbuf + (p / kPieces) * kRowBytes + (p % kPieces) * 16) = r[j];
        }
    };

    int tile = blockIdx.x;
    if (tile >= dm.tiles) return;
    u32x4 stage[kLoads];
    fetch(tile, stage);

    for (;;) {
        park(stage);
        __syncthreads();  // the tile is in LDS
        const int next = tile + gridDim.x;
        const bool more = next < dm.tiles;
        if (more) fetch(next, stage);  // in flight during the MFMAs and the stores below

        if (has_cols) {
#pragma unroll
            for (int half = 0; half < kRows / 32; ++half) {  // 32 rows at a time through the same accumulators
                f32x16 acc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_tab + (kg * 2 + t) * 16 + 4 * q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = b4[i];
                    }
#pragma unroll
                for (int s = 0; s < kSteps; ++s) {
                    // X fragment: X[tile row 32 half + nl][16 s + 8 kg .. + 8)
                    const u32x4 a = *reinterpret_cast<const u32x4*>(xbuf + (32 * half + nl) * kRowBytes + (16 * s + 8 * kg) * 2);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wreg[0][s]), as_bf16x8(a), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wreg[1][s]), as_bf16x8(a), acc[1], 0, 0, 0);
                }
                // lane = tile row 32 half + nl; registers 4 q .. 4 q + 3 = columns 32 t + 8 q + 4 kg .. + 3 -> bf16 -> LDS [row][col]
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v0 = acc[t][4 * q], v1 = acc[t][4 * q + 1], v2 = acc[t][4 * q + 2], v3 = acc[t][4 * q + 3];
                        if (RELU && !HAS_RES) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        *reinterpret_cast<u32x2*>(obuf + (32 * half + nl) * kOutStride + (32 * t + 8 * q + 4 * kg) * 2) =
                            u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (HM) {
                // per head (32 columns = 64 B per row): 4 lanes x 16 B per row, 16 consecutive rows = 1 KB per store instruction
                const unsigned char* mask = reinterpret_cast<const unsigned char*>(R);
                const int heads = dm.N / 32;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int pass = 0; pass < kRows / 16; ++pass) {
                        const int row = pass * 16 + (lane >> 2);
                        const long grow = (long)tile * kRows + row;
                        if (grow < dm.M) {
                            u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + hh * 64 + (lane & 3) * 16);
                            if (mask != nullptr && mask[grow]) v = u32x4{0u, 0u, 0u, 0u};
                            const long nb = grow / dm.S, sp = grow - nb * dm.S;
                            const int head = col0 / 32 + hh;
                            *reinterpret_cast<u32x4*>(Y + ((nb * heads + head) * dm.S + sp) * 32 + (lane & 3) * 8) = v;
                        }
                    }
                }
            } else
            // rows leave as whole 128-byte lines: 8 lanes x 16 B per row, 8 rows per store instruction
#pragma unroll
            for (int pass = 0; pass < kRows / 8; ++pass) {
                const int row = pass * 8 + (lane >> 3);
                const long grow = (long)tile * kRows + row;
                u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + (lane & 7) * 16);
                if (grow < dm.M) {
                    if constexpr (HAS_RES) {  // + identity (same coordinates as y), then the activation
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
        if (!more) break;
        __syncthreads();  // every wave has finished reading the tile: it may be overwritten
        tile = next;
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

namespace {
template <int K, bool RELU, bool HAS_RES, bool HM = false>
int launch_shortk(const void* x, const void* weight, const void* bias, const void* residual, void* y, long M, int N,
                  hipStream_t stream, int S = 0, const int* gather = nullptr) {
    GemmDims dm;
    dm.M = M; dm.N = N; dm.tiles = (int)((M + kRows - 1) / kRows); dm.S = S;
    dm.gs = gather ? gather[0] : 1;
    dm.gWo = gather ? gather[2] : 1; dm.gHoWo = gather ? gather[1] * gather[2] : 1;
    dm.gW = gather ? gather[4] : 1; dm.gHW = gather ? gather[3] * gather[4] : 1;
    const size_t lds = kRows * (K * 2 + 16) + 4 * kRows * kOutStride + 4 * 64 * sizeof(float);
    const int cols = (N + 255) / 256;
    int gx = 512 / cols;  // persistent: about two workgroups per CU in total
    if (gx > dm.tiles) gx = dm.tiles;
    if (gx < 1) gx = 1;
    void* args[] = {&x, &weight, &bias, &residual, &y, &dm};
    static unsigned long long attr_done = 0;  // per instantiation (lds depends on K only), one bit per device
    (void)ensure_dynamic_lds(reinterpret_cast<const void*>(linear_shortk_kernel<K, RELU, HAS_RES, HM>), (int)lds, &attr_done);
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(linear_shortk_kernel<K, RELU, HAS_RES, HM>), dim3(gx, cols), dim3(256), args,
                                   lds, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_linear_shortk: %s", hipGetErrorString(e));
    return check_launch("alo_linear_shortk");
}
}  // namespace

extern "C" int alo_linear_shortk(const void* x, const void* weight, const void* bias, const void* residual, void* y, long M,
                                 int N, int K, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && weight && y, ALO_ERR_INVALID_ARGUMENT, "alo_linear_shortk: null pointer argument");
    ALO_REQUIRE(M > 0 && N > 0 && N % 64 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_linear_shortk: M must be positive and N a positive multiple of 64 (M=%ld N=%d)", M, N);
    ALO_REQUIRE(K == 64 || K == 128 || K == 256, ALO_ERR_UNSUPPORTED, "alo_linear_shortk: K must be 64, 128 or 256, got %d", K);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_linear_shortk: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_linear_shortk: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define ALO_GEMM_CASE(KK)                                                                                       \
    if (K == KK) {                                                                                               \
        if (residual) return relu ? launch_shortk<KK, true, true>(x, weight, bias, residual, y, M, N, st)       \
                                  : launch_shortk<KK, false, true>(x, weight, bias, residual, y, M, N, st);     \
        return relu ? launch_shortk<KK, true, false>(x, weight, bias, residual, y, M, N, st)                    \
                    : launch_shortk<KK, false, false>(x, weight, bias, residual, y, M, N, st);                  \
    }
    ALO_GEMM_CASE(64) ALO_GEMM_CASE(128) ALO_GEMM_CASE(256)
#undef ALO_GEMM_CASE
    return ALO_ERR_UNSUPPORTED;
}

// 1x1 convolution with a spatial stride over an NHWC map, resident-weight flavour: the kept pixels are addressed by the tile
// loader itself (no gathered copy of the input).  Declared in alo_hotpath.h as part of alo_conv1x1_nhwc (gemm_packed.hip).
extern "C" int alo_internal_shortk_gather(const void* x, const void* weight, const void* bias, const void* residual, void* y,
                                          long M, int N, int K, int relu, const int* gather, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
#define ALO_GEMM_CASE(KK)                                                                                                  \
    if (K == KK) {                                                                                                          \
        if (residual) return relu ? launch_shortk<KK, true, true>(x, weight, bias, residual, y, M, N, st, 0, gather)       \
                                  : launch_shortk<KK, false, true>(x, weight, bias, residual, y, M, N, st, 0, gather);     \
        return relu ? launch_shortk<KK, true, false>(x, weight, bias, residual, y, M, N, st, 0, gather)                    \
                    : launch_shortk<KK, false, false>(x, weight, bias, residual, y, M, N, st, 0, gather);                  \
    }
    ALO_GEMM_CASE(64) ALO_GEMM_CASE(128) ALO_GEMM_CASE(256)
#undef ALO_GEMM_CASE
    return fail(ALO_ERR_UNSUPPORTED, "alo_conv1x1_nhwc: the resident-weight kernel needs Cin in (64, 128, 256), got %d", K);
}

// ------------------------------------------------------------------------------------------------------------------
// The transformer layer's feed-forward block in one kernel:  y = relu(x W1^T + b1) W2^T + b2,  x, y (M, 256), hidden F.
// W1 / W2 arrive PACKED in MFMA fragment order (alo_pack_mfma_b: [row tile of 32][k step of 16][lane = 32 kg + n][8]).
//
// Run as two library GEMMs the (M, F) hidden activation makes a round trip through HBM (2 x 364 MB at M = 177784, F = 1024)
// and the pair takes 316 us.  Here a workgroup owns 64 rows: it keeps them in LDS, produces the hidden activation 256
// units at a time (each wave 64 of them) straight into LDS as bf16, and immediately contracts that chunk into its
// 64 x 256 output accumulators (each wave 64 output columns), so the hidden tensor never exists in memory.  The weights
// (2 x 512 KB, L2 resident) stream through registers; x is read once, y written once, in whole lines.
// ------------------------------------------------------------------------------------------------------------------
namespace alo {
namespace {

constexpr int kFfnRows = 64;
constexpr int kFfnStride = 256 * 2 + 16;  // LDS row stride of the 64 x 256 bf16 tiles (x, hidden chunk, output staging)

struct FfnDims {
    long M;
    int F;      // hidden width, multiple of 256
    int tiles;  // ceil(M / 64)
};

__global__ void __launch_bounds__(256, 2)
ffn256_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1, const bf16_t* __restrict__ B1,
              const bf16_t* __restrict__ W2, const bf16_t* __restrict__ B2, bf16_t* __restrict__ Y, const FfnDims dm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xs = smem;                            // x tile        [64][256] bf16
    unsigned char* const hs = smem + kFfnRows * kFfnStride;    // hidden chunk  [64][256] bf16, then the output staging
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nl = lane & 31, kg = lane >> 5;
    const int F = dm.F;

    // biases as fp32 in LDS (b1: F values, b2: 256); the products are computed transposed (weights = the MFMA's row operand),
    // so a lane owns ONE row of the tile and four consecutive columns per accumulator quad: bias, activation and the bf16
    // conversion work on quads and leave as 8-byte LDS writes
    float* const b1s = reinterpret_cast<float*>(smem + 2 * kFfnRows * kFfnStride);
    float* const b2s = b1s + F;
    for (int i = tid; i < F; i += 256) b1s[i] = B1 ? bf16_to_f32(B1[i].bits) : 0.f;
    b2s[tid] = B2 ? bf16_to_f32(B2[tid].bits) : 0.f;

    // Weight fragments arrive in batches of KB k-steps (2 column tiles x KB x 16 B per lane) through two register buffers:
    // batch i+1 is requested before batch i is consumed, across phase and round boundaries too (L2 latency ~ the MFMA time
    // of one batch).  sched_barrier keeps the compiler from sinking the requests next to their first use.
    constexpr int KB = 2;            // k-steps per batch
    constexpr int NBATCH = 16 / KB;  // batches per phase
    auto load_batch = [&](u32x4 (&buf)[2][KB], const bf16_t* frag0, size_t tile_stride, int batch) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < KB; ++j)
                buf[t][j] = *reinterpret_cast<const u32x4*>(frag0 + t * tile_stride + (size_t)(KB * batch + j) * 512);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_batch = [&](f32x16 (&acc)[2][2], const unsigned char* a_lds, const u32x4 (&buf)[2][KB], int batch) {
        u32x4 af[2][KB];  // A fragments of the whole batch first: their LDS latency overlaps instead of preceding each MFMA group
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int s = KB * batch + j;
            af[0][j] = *reinterpret_cast<const u32x4*>(a_lds + nl * kFfnStride + (16 * s + 8 * kg) * 2);
            af[1][j] = *reinterpret_cast<const u32x4*>(a_lds + (32 + nl) * kFfnStride + (16 * s + 8 * kg) * 2);
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const u32x4 a0 = af[0][j], a1 = af[1][j];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(buf[0][j]), as_bf16x8(a0), acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(buf[1][j]), as_bf16x8(a0), acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(buf[0][j]), as_bf16x8(a1), acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(buf[1][j]), as_bf16x8(a1), acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int fs = F / 16;  // k steps per output-column tile of the packed W2
    // packed fragment (row tile, k step) = 64 lanes x 16 B contiguous: a load instruction reads 8 whole lines
    auto w1_frag = [&](int r0) { return W1 + ((size_t)((r0 + 64 * wave) / 32) * 16 * 64 + lane) * 8; };  // tile stride 16 * 512
    auto w2_frag = [&](int r0) { return W2 + (((size_t)(2 * wave) * fs + r0 / 16) * 64 + lane) * 8; };     // tile stride fs * 512

    for (int tile = blockIdx.x; tile < dm.tiles; tile += gridDim.x) {
        // ---- x tile -> LDS (64 rows x 512 B, 8 pieces of 16 B per thread, rows contiguous across lanes); rows past the end are
        // read from the last row (never stored) so that all eight requests go out back to back -----------------------------
        {
            u32x4 xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = tid + 256 * j;
                long row = (long)tile * kFfnRows + (p >> 5);
                row = row < dm.M ? row : dm.M - 1;
                xv[j] = *reinterpret_cast<const u32x4*>(X + row * 256 + (p & 31) * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = tid + 256 * j;
                *reinterpret_cast<u32x4*>(xs + (p >> 5) * kFfnStride + (p & 31) * 16) = xv[j];
            }
        }
        u32x4 bufa[2][KB], bufb[2][KB];
        load_batch(bufa, w1_frag(0), (size_t)16 * 512, 0);
        __syncthreads();

        f32x16 acc2[2][2];  // [row tile][column tile] of this wave's 64 output columns
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[a][t][i] = 0.f;

        for (int r0 = 0; r0 < F; r0 += 256) {
            // ---- phase 1: this wave's 64 hidden units of the chunk: h = relu(x W1[r0 + 64 wave + ...]^T + b1) ----------------
            f32x16 acc1[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc1[a][t][i] = 0.f;
            const bf16_t* w1p = w1_frag(r0);
            const bf16_t* w2p = w2_frag(r0);
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt += 2) {
                load_batch(bufb, w1p, (size_t)16 * 512, bt + 1);
                mma_batch(acc1, xs, bufa, bt);
                if (bt + 2 < NBATCH) load_batch(bufa, w1p, (size_t)16 * 512, bt + 2);
                else load_batch(bufa, w2p, (size_t)fs * 512, 0);
                mma_batch(acc1, xs, bufb, bt + 1);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 64 * wave + 32 * t + 8 * q + 4 * kg;  // registers 4 q .. 4 q + 3 = hidden units col .. col + 3
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + r0 + col);
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const float v0 = fmaxf(acc1[a][t][4 * q] + bb[0], 0.f), v1 = fmaxf(acc1[a][t][4 * q + 1] + bb[1], 0.f);
                        const float v2 = fmaxf(acc1[a][t][4 * q + 2] + bb[2], 0.f), v3 = fmaxf(acc1[a][t][4 * q + 3] + bb[3], 0.f);
                        *reinterpret_cast<u32x2*>(hs + (32 * a + nl) * kFfnStride + col * 2) = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    }
                }
            __syncthreads();  // the whole 64 x 256 hidden chunk is in LDS

            // ---- phase 2: out[:, 64 wave ..] += h_chunk (64 x 256) W2[64 wave + ..][r0 .. r0 + 256)^T ------------------------
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt += 2) {
                load_batch(bufb, w2p, (size_t)fs * 512, bt + 1);
                mma_batch(acc2, hs, bufa, bt);
                if (bt + 2 < NBATCH) load_batch(bufa, w2p, (size_t)fs * 512, bt + 2);
                else if (r0 + 256 < F) load_batch(bufa, w1_frag(r0 + 256), (size_t)16 * 512, 0);
                mma_batch(acc2, hs, bufb, bt + 1);
            }
            __syncthreads();  // everyone is done reading the chunk before it is overwritten
        }

        // ---- epilogue: + b2 -> bf16 -> staging (the hidden-chunk buffer) -> whole-line stores ---------------------------------
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 64 * wave + 32 * t + 8 * q + 4 * kg;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + col);
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    *reinterpret_cast<u32x2*>(hs + (32 * a + nl) * kFfnStride + col * 2) =
                        u32x2{pack_bf16x2(acc2[a][t][4 * q] + bb[0], acc2[a][t][4 * q + 1] + bb[1]),
                              pack_bf16x2(acc2[a][t][4 * q + 2] + bb[2], acc2[a][t][4 * q + 3] + bb[3])};
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = tid + 256 * j;
            const long row = (long)tile * kFfnRows + (p >> 5);
            if (row < dm.M)
                *reinterpret_cast<u32x4*>(Y + row * 256 + (p & 31) * 8) =
                    *reinterpret_cast<const u32x4*>(hs + (p >> 5) * kFfnStride + (p & 31) * 16);
        }
        __syncthreads();  // staging and x tile are rewritten by the next tile
    }
}

}  // namespace
}  // namespace alo

namespace alo {
namespace {
// W (N, K) row-major -> fragments [N / 32][K / 16][64 lanes][8]: lane (kg, n) of fragment (t, s) holds W[32 t + n][16 s + 8 kg ..+8)
__global__ void __launch_bounds__(256)
pack_mfma_b_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ P, int N, int K) {
    const long total = (long)N * K / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i % 64);
        const long frag = i / 64;
        const int s = (int)(frag % (K / 16)), t = (int)(frag / (K / 16));
        const int n = 32 * t + (lane & 31), k = 16 * s + 8 * (lane >> 5);
        *reinterpret_cast<u32x4*>(P + i * 8) = *reinterpret_cast<const u32x4*>(W + (size_t)n * K + k);
    }
}
}  // namespace
}  // namespace alo

extern "C" int alo_pack_mfma_b(const void* w, void* packed, int N, int K, int dtype, void* stream) {
    ALO_REQUIRE(w && packed, ALO_ERR_INVALID_ARGUMENT, "alo_pack_mfma_b: null pointer argument");
    ALO_REQUIRE(N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_pack_mfma_b: N must be a multiple of 32 and K a multiple of 16 (N=%d K=%d)", N, K);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_pack_mfma_b: bf16 only (dtype %d)", dtype);
    long total = (long)N * K / 8;
    unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    void* args[] = {&w, &packed, &N, &K};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(pack_mfma_b_kernel), dim3(blocks), dim3(256), args, 0,
                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_pack_mfma_b: %s", hipGetErrorString(e));
    return check_launch("alo_pack_mfma_b");
}

extern "C" int alo_ffn256(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* y, long M,
                          int F, int dtype, void* stream) {
    ALO_REQUIRE(x && w1 && w2 && y, ALO_ERR_INVALID_ARGUMENT, "alo_ffn256: null pointer argument");
    ALO_REQUIRE(M > 0 && F > 0 && F % 256 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_ffn256: M must be positive and the hidden width a positive multiple of 256 (M=%ld F=%d)", M, F);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_ffn256: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)y) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_ffn256: pointers must be 16-byte aligned");
    FfnDims dm;
    dm.M = M; dm.F = F; dm.tiles = (int)((M + kFfnRows - 1) / kFfnRows);
    const size_t lds = 2 * kFfnRows * kFfnStride + ((size_t)F + 256) * sizeof(float);
    int gx = dm.tiles < 512 ? dm.tiles : 512;
    void* args[] = {&x, &w1, &b1, &w2, &b2, &y, &dm};
    if (lds > 160 * 1024) return fail(ALO_ERR_UNSUPPORTED, "alo_ffn256: hidden width %d needs %zu bytes of LDS", F, lds);
    {   // lds grows with F: keep the largest limit set so far per device (a process-wide flag would miss a second device or a wider F)
        static int limit_set[64] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if ((int)lds > __atomic_load_n(&limit_set[dev & 63], __ATOMIC_ACQUIRE)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
                __atomic_store_n(&limit_set[dev & 63], (int)lds, __ATOMIC_RELEASE);
        }
    }
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(ffn256_kernel), dim3(gx), dim3(256), args, lds,
                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_ffn256: %s", hipGetErrorString(e));
    return check_launch("alo_ffn256");
}

extern "C" int alo_value_proj_head_major(const void* x, const void* weight, const void* bias, const void* padding_mask,
                                         void* value_hm, int batch, int S, int heads, int K, int dtype, void* stream) {
    ALO_REQUIRE(x && weight && value_hm, ALO_ERR_INVALID_ARGUMENT, "alo_value_proj_head_major: null pointer argument");
    ALO_REQUIRE(batch > 0 && S > 0 && heads > 0 && heads % 2 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_value_proj_head_major: batch, S must be positive and the head count even (batch=%d S=%d heads=%d)", batch, S,
                heads);
    ALO_REQUIRE(K == 64 || K == 128 || K == 256, ALO_ERR_UNSUPPORTED, "alo_value_proj_head_major: K must be 64, 128 or 256, got %d", K);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_value_proj_head_major: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)value_hm) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_value_proj_head_major: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long M = (long)batch * S;
    const int N = heads * 32;
    if (K == 64) return launch_shortk<64, false, false, true>(x, weight, bias, padding_mask, value_hm, M, N, st, S);
    if (K == 128) return launch_shortk<128, false, false, true>(x, weight, bias, padding_mask, value_hm, M, N, st, S);
    return launch_shortk<256, false, false, true>(x, weight, bias, padding_mask, value_hm, M, N, st, S);
}
