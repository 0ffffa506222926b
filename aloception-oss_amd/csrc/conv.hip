// 3x3 convolution, padding 1, stride 1 or 2, NHWC bf16 (fp32 accumulation) with bias + ReLU in the epilogue: the middle
// convolution of the ResNet bottlenecks (alonet/detr/backbone.py + torchvision Bottleneck.conv2, FrozenBatchNorm folded in).
//
// An implicit GEMM on v_mfma_f32_32x32x16_bf16: y[p, :] = sum over the 9 taps of x[pixel of tap(p), :] W_tap^T, K = 9 * Cin.
//   * a workgroup (two waves) owns 64 consecutive output pixels of one image (flattened y * Wo + x, so rows of any width
//     tile without waste) and 128 output channels, 64 per wave = 2 x 2 MFMA tiles per wave;
//   * the input pixels the tile touches are copied, CHUNK channels at a time, into LDS with whole-line loads that are issued
//     one chunk ahead of their use (register staged), so their round trip hides behind the previous chunk's MFMAs:
//       stride 1: three runs of 66 consecutive input pixels (rows y-1, y, y+1; flattened index p - 1 .. p + 64), tap (dy, dx)
//                 of tile pixel m reads slot 66 dy + dx + m;
//       stride 2: per tap row dy a run of 65 odd-column and a run of 64 even-column pixels of input row 2 y + dy - 1 — in the
//                 four parity phases of the input a stride-2 tap is again a shift of the flattened OUTPUT index;
//     A fragments are ds_read from there and zeroed per lane where the tap falls outside the image (the zero padding);
//   * the weights arrive pre-packed in MFMA B-fragment order ([Cout / 32][9 Cin / 16][64 lanes][8], k = tap * Cin + c), are
//     read as whole 1 KB lines per instruction and stream through two register buffers, one batch of two k-steps ahead;
//   * Cout == 64 (ResNet layer1) would leave the second wave without columns: there the two waves split K instead (each takes
//     every other weight batch) and the partial sums meet in LDS.
#include "common.hpp"

namespace alo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

constexpr int kPix = 64;                     // output pixels per tile
constexpr int kThreads = 128;                // two waves
constexpr int kOutStride = 64 * 2 + 16;      // LDS bytes per pixel of a wave's 64 x 64 output block
constexpr int kRedOffset = kPix * kOutStride;  // K-split: wave 1's fp32 partial sums [64 regs][64 lanes] sit behind wave 0's output block

template <int STRIDE>
struct Geo {
    static constexpr int kChunk = STRIDE == 1 ? 64 : 32;     // input channels staged at once
    static constexpr int kBpt = kChunk / 32;                 // weight batches (two k-steps each) per tap and chunk
    static constexpr int kRun = STRIDE == 1 ? kPix + 2 : 2 * kPix + 1;  // staged pixels per tap row
    static constexpr int kSlots = 3 * kRun;
    static constexpr int kPixStride = kChunk * 2 + 16;       // LDS bytes per staged pixel (+16: conflict-free 16-byte reads)
    static constexpr int kPieces = kSlots * (kChunk / 8);    // 16-byte pieces per chunk
    static constexpr int kIters = (kPieces + kThreads - 1) / kThreads;
    static constexpr int kLds = kSlots * kPixStride;
    static_assert(kLds >= kRedOffset + 64 * 64 * 4, "the epilogue's staging aliases the halo");
};

struct ConvDims {
    int N, H, W, Ho, Wo, Cin, Cout;
    int tiles_per_image;
    int relu;
    int zsplit, cin_per_z;   // split-K over gridDim.z: workgroup z reduces input channels [z * cin_per_z, + cin_per_z)
};

struct WFrag { u32x4 v[2][2]; };    // [k-step of the batch][column tile]

// flattened input pixel (clamped into the image) behind staging slot `slot` of the tile whose first output pixel is p0
template <int STRIDE>
__device__ __forceinline__ int slot_pixel(int slot, int p0, const ConvDims& dm) {
    if (STRIDE == 1) {
        const int run = slot / Geo<1>::kRun, pix = slot - run * Geo<1>::kRun;
        const int q = p0 + (run - 1) * dm.W - 1 + pix;
        const int last = dm.H * dm.W - 1;
        return q < 0 ? 0 : (q > last ? last : q);
    }
    const int dy = slot / Geo<2>::kRun, r = slot - dy * Geo<2>::kRun;
    const int odd = r < kPix + 1 ? 1 : 0;                        // column parity of this run
    int f = p0 + (odd ? r - 1 : r - (kPix + 1)) - (dy == 0 ? dm.Wo : 0);   // output-index space: (yo + oy) * Wo + xo + ox
    const int last = dm.Ho * dm.Wo - 1;
    f = f < 0 ? 0 : (f > last ? last : f);
    const int yp = f / dm.Wo, xp = f - yp * dm.Wo;
    int iy = 2 * yp + (dy == 1 ? 0 : 1), ix = 2 * xp + odd;
    iy = iy > dm.H - 1 ? dm.H - 1 : iy;
    ix = ix > dm.W - 1 ? dm.W - 1 : ix;
    return iy * dm.W + ix;
}

template <int STRIDE, bool KSPLIT>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ bias,
               bf16_t* __restrict__ Y, float* __restrict__ partial, const ConvDims dm) {
    using G = Geo<STRIDE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nl = lane & 31, kg = lane >> 5;
    const int HWo = dm.Ho * dm.Wo;
    const int col0 = KSPLIT ? blockIdx.y * 64 : blockIdx.y * 128 + wave * 64;  // this wave's 64 output channels
    const bool has_cols = col0 < dm.Cout;                               // Cout % 64 == 0
    float* const bias_s = reinterpret_cast<float*>(smem + G::kLds);     // the workgroup's 128 (K-split: 64) bias values, fp32
    {
        const int c = (KSPLIT ? blockIdx.y * 64 : blockIdx.y * 128) + tid;
        bias_s[tid] = (bias != nullptr && c < dm.Cout && (!KSPLIT || tid < 64)) ? bf16_to_f32(bias[c].bits) : 0.f;
    }

    // Persistent workgroups.  Workgroup ids go round-robin over the 8 XCDs, so XCD x (ids = x mod 8) takes the x-th eighth of
    // the tiles: neighbouring tiles — which share two of their three halo rows — meet in the same L2.
    const int ntiles = dm.tiles_per_image * dm.N;
    const int per_xcd = (ntiles + 7) >> 3, xcd = blockIdx.x & 7, lanes_per_xcd = gridDim.x >> 3;
    const int tile_end = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
    int tile = xcd * per_xcd + (blockIdx.x >> 3);
    if (tile >= tile_end) return;

    const int ksteps_per_tap = dm.Cin / 16;
    const size_t ctile_stride = (size_t)9 * ksteps_per_tap * 512;      // elements between the packed column tiles
    const bf16_t* wfrag = Wp + (size_t)(col0 / 32) * ctile_stride + lane * 8;

    // this wave's weight batches of a chunk: all 9 kBpt of them, or every other one when the two waves split K
    constexpr int kBatches = 9 * G::kBpt;
    const int my_batches = KSPLIT ? (kBatches - wave + 1) / 2 : kBatches;
    auto batch = [&](int i) { return KSPLIT ? wave + 2 * i : i; };

    // halo pieces travel through registers and are requested one (tile, chunk) AHEAD of their use
    u32x4 stage[G::kIters];
    auto load_halo = [&](int tl, int c0) {
        const int im = tl / dm.tiles_per_image, first = (tl - im * dm.tiles_per_image) * kPix;
        const bf16_t* ximg = X + (size_t)im * dm.H * dm.W * dm.Cin + c0;
#pragma unroll
        for (int it = 0; it < G::kIters; ++it) {
            const int i = tid + it * kThreads;
            const int piece = i % (G::kChunk / 8), slot = i / (G::kChunk / 8);
            if (i < G::kPieces)
                stage[it] = *reinterpret_cast<const u32x4*>(ximg + (size_t)slot_pixel<STRIDE>(slot, first, dm) * dm.Cin + piece * 8);
        }
    };
    const int cbeg = blockIdx.z * dm.cin_per_z, cend = cbeg + dm.cin_per_z;
    load_halo(tile, cbeg);

    for (; tile < tile_end; tile += lanes_per_xcd) {
        const int img = tile / dm.tiles_per_image;
        const int p0 = (tile - img * dm.tiles_per_image) * kPix;       // first output pixel of the tile inside the image

        // which of the 9 taps exist for this lane's two pixels (row tiles a = 0, 1: output pixel p0 + 32 a + nl)
        unsigned tapmask[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int q = p0 + 32 * a + nl;
            const int y = q / dm.Wo, x = q - y * dm.Wo;
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y * STRIDE + t / 3 - 1, xx = x * STRIDE + t % 3 - 1;
                if (q < HWo && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) m |= 1u << t;
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

        for (int c0 = cbeg; c0 < cend; c0 += G::kChunk) {
            // weights of batch kb of this chunk: tap kb / kBpt, k-steps 2 (kb % kBpt) and + 1 of the chunk
            auto load_w = [&](WFrag& f, int kb) {
                const bf16_t* p = wfrag + (size_t)((kb / G::kBpt) * ksteps_per_tap + c0 / 16 + 2 * (kb % G::kBpt)) * 512;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        f.v[j][b] = *reinterpret_cast<const u32x4*>(p + (size_t)j * 512 + b * ctile_stride);
            };
            auto compute = [&](const WFrag& f, int kb) {
                const int t = kb / G::kBpt, dy = t / 3, dx = t - 3 * dy;
                const int slot = STRIDE == 1 ? dy * G::kRun + dx : dy * G::kRun + (dx == 0 ? 0 : (dx == 1 ? kPix + 1 : 1));
                const bool ok0 = (tapmask[0] >> t) & 1u, ok1 = (tapmask[1] >> t) & 1u;
                const unsigned char* a_base = halo + (slot + nl) * G::kPixStride + kg * 16 + (kb % G::kBpt) * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 a0 = *reinterpret_cast<const u32x4*>(a_base + j * 32);
                    u32x4 a1 = *reinterpret_cast<const u32x4*>(a_base + 32 * G::kPixStride + j * 32);
                    if (!ok0) a0 = u32x4{0u, 0u, 0u, 0u};
                    if (!ok1) a1 = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(f.v[j][b]), as_bf16x8(a0), acc[0][b], 0, 0, 0);
                        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(f.v[j][b]), as_bf16x8(a1), acc[1][b], 0, 0, 0);
                    }
                }
            };
            WFrag w0, w1;
            if (has_cols) load_w(w0, batch(0));

#pragma unroll
            for (int it = 0; it < G::kIters; ++it) {
                const int i = tid + it * kThreads;
                const int piece = i % (G::kChunk / 8), slot = i / (G::kChunk / 8);
                if (i < G::kPieces) *reinterpret_cast<u32x4*>(halo + slot * G::kPixStride + piece * 16) = stage[it];
            }
            __syncthreads();
            if (c0 + G::kChunk < cend) load_halo(tile, c0 + G::kChunk);
            else if (tile + lanes_per_xcd < tile_end) load_halo(tile + lanes_per_xcd, cbeg);

            if (has_cols) {
#pragma unroll 1
                for (int i = 0; i < my_batches; i += 2) {
                    if (i + 1 < my_batches) load_w(w1, batch(i + 1));
                    compute(w0, batch(i));
                    if (i + 2 < my_batches) load_w(w0, batch(i + 2));
                    if (i + 1 < my_batches) compute(w1, batch(i + 1));
                }
            }
            __syncthreads();  // the halo is overwritten by the next channel chunk (and by the epilogue's staging)
        }

        if (KSPLIT) {  // wave 1 hands its partial sums to wave 0
            float* red = reinterpret_cast<float*>(smem + kRedOffset);
            if (wave == 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * 2 + b) * 16 + r) * 64 + lane];
            }
        }

        if (!KSPLIT && dm.zsplit > 1) {
            // split-K: the raw fp32 partial sums of this channel range go to partial[z][pixel][channel]; conv_splitk_finalize_kernel
            // adds the ranges up in a fixed order (deterministic), then bias / ReLU / bf16
            if (has_cols) {
                float* pz = partial + ((size_t)blockIdx.z * dm.N + img) * HWo * dm.Cout + col0;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int q = p0 + 32 * a + nl;
                    if (q < HWo) {
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd)
                                *reinterpret_cast<f32x4*>(pz + (size_t)q * dm.Cout + 32 * b + 8 * qd + 4 * kg) =
                                    f32x4{acc[a][b][4 * qd], acc[a][b][4 * qd + 1], acc[a][b][4 * qd + 2], acc[a][b][4 * qd + 3]};
                    }
                }
            }
        } else if (has_cols && !(KSPLIT && wave == 1)) {
            unsigned char* const obuf = smem + (KSPLIT ? 0 : wave) * (kPix * kOutStride);
            // the product is computed transposed (weights = the MFMA's row operand): lane = output pixel 32 a + nl, registers
            // 4 q .. 4 q + 3 = channels col0 + 32 b + 8 q + 4 kg .. + 3 -> bias, ReLU, bf16, one 8-byte LDS write per quad
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * b + 8 * q + 4 * kg;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_s + (KSPLIT ? 0 : wave) * 64 + c);
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float v0 = acc[a][b][4 * q] + bb[0], v1 = acc[a][b][4 * q + 1] + bb[1];
                        float v2 = acc[a][b][4 * q + 2] + bb[2], v3 = acc[a][b][4 * q + 3] + bb[3];
                        if (dm.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        *reinterpret_cast<u32x2*>(obuf + (32 * a + nl) * kOutStride + c * 2) = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
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
                if (q < HWo)
                    *reinterpret_cast<u32x4*>(Y + ((size_t)img * HWo + q) * dm.Cout + col0 + (lane & 7) * 8) =
                        *reinterpret_cast<const u32x4*>(obuf + row * kOutStride + (lane & 7) * 16);
            }
        }
        __syncthreads();  // the staging is read out before the next tile's halo lands on it
    }
}

// y[p][c] = act(sum_z partial[z][p][c] + bias[c]) -> bf16; one thread per 8 channels
__global__ void __launch_bounds__(256)
conv_splitk_finalize_kernel(const float* __restrict__ partial, const bf16_t* __restrict__ bias, bf16_t* __restrict__ Y, long rows,
                            int Cout, int Z, int relu) {
    const long n8 = rows * (Cout / 8);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        const int c = (int)(e % Cout);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bias ? bf16_to_f32(bias[c + j].bits) : 0.f;
        for (int z = 0; z < Z; ++z) {
            const float* p = partial + (size_t)z * rows * Cout + e;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
        }
        if (relu)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        *reinterpret_cast<u32x4*>(Y + e) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
}

// split-K factor: only when the tiles alone leave most of the chip idle (the 2048 -> 256 input projection: 80 workgroups)
int conv_zsplit(int ntiles, int Cin, int Cout) {
    if (Cout == 64) return 1;
    const int base = ntiles * ((Cout + 127) / 128);
    int z = 1;
    while (base * z * 2 <= 1024 && Cin % (z * 2 * 64) == 0 && Cin / (z * 2) >= 128) z *= 2;
    return z;
}

template <int STRIDE, bool KSPLIT>
hipError_t launch_conv(const void* x, const void* w, const void* bias, void* y, void* partial, const ConvDims& dm, hipStream_t stream) {
    const void* kern = reinterpret_cast<const void*>(conv3x3_kernel<STRIDE, KSPLIT>);
    constexpr int lds = Geo<STRIDE>::kLds + kThreads * 4;
    void* args[] = {&x, &w, &bias, &y, &partial, const_cast<ConvDims*>(&dm)};
    const unsigned gy = KSPLIT ? dm.Cout / 64 : (dm.Cout + 127) / 128;
    const int ntiles = dm.tiles_per_image * dm.N;
    int per_xcd = (ntiles + 7) / 8;
    if (per_xcd > 128) per_xcd = 128;   // 32 CUs per XCD x up to 4 resident workgroups
    return hipLaunchKernel(kern, dim3((unsigned)(8 * per_xcd), gy, (unsigned)dm.zsplit), dim3(kThreads), args, lds, stream);
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" size_t alo_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
    const long hwo = (long)((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    const int z = conv_zsplit((int)((hwo + kPix - 1) / kPix) * N, Cin, Cout);
    return z > 1 ? (size_t)z * N * hwo * Cout * sizeof(float) : 0;
}

extern "C" int alo_conv3x3_nhwc(const void* x, const void* w_packed, const void* bias, void* y, void* workspace, int N, int H,
                                int W, int Cin, int Cout, int stride, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && w_packed && y, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_nhwc: null pointer argument");
    ALO_REQUIRE(N > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_nhwc: N, H, W must be positive");
    ALO_REQUIRE(stride == 1 || stride == 2, ALO_ERR_UNSUPPORTED, "alo_conv3x3_nhwc: stride must be 1 or 2 (got %d)", stride);
    ALO_REQUIRE(Cin >= 64 && Cin % 64 == 0 && Cout >= 64 && Cout % 64 == 0, ALO_ERR_UNSUPPORTED,
                "alo_conv3x3_nhwc: Cin and Cout must be multiples of 64 (Cin=%d Cout=%d)", Cin, Cout);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_conv3x3_nhwc: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_conv3x3_nhwc: pointers must be 16-byte aligned");
    ALO_REQUIRE((long)H * W < (1L << 24) && (long)N * H * W * (long)(Cin > Cout ? Cin : Cout) < (1L << 40), ALO_ERR_UNSUPPORTED,
                "alo_conv3x3_nhwc: image too large");
    ConvDims dm;
    dm.N = N; dm.H = H; dm.W = W; dm.Cin = Cin; dm.Cout = Cout; dm.relu = relu;
    dm.Ho = (H - 1) / stride + 1;
    dm.Wo = (W - 1) / stride + 1;
    dm.tiles_per_image = (dm.Ho * dm.Wo + kPix - 1) / kPix;
    dm.zsplit = workspace ? conv_zsplit(dm.tiles_per_image * N, Cin, Cout) : 1;   // no workspace: no split-K
    dm.cin_per_z = Cin / dm.zsplit;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool ksplit = Cout == 64;
    hipError_t e = stride == 1 ? (ksplit ? launch_conv<1, true>(x, w_packed, bias, y, workspace, dm, s) : launch_conv<1, false>(x, w_packed, bias, y, workspace, dm, s))
                               : (ksplit ? launch_conv<2, true>(x, w_packed, bias, y, workspace, dm, s) : launch_conv<2, false>(x, w_packed, bias, y, workspace, dm, s));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_conv3x3_nhwc: %s", hipGetErrorString(e));
    if (dm.zsplit > 1) {
        const long rows = (long)N * dm.Ho * dm.Wo;
        const float* part = static_cast<const float*>(workspace);
        int z = dm.zsplit;
        long blocks = (rows * (Cout / 8) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        void* args[] = {&part, &bias, &y, const_cast<long*>(&rows), &Cout, &z, &relu};
        e = hipLaunchKernel(reinterpret_cast<const void*>(conv_splitk_finalize_kernel), dim3((unsigned)blocks), dim3(256), args, 0, s);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_conv3x3_nhwc (finalize): %s", hipGetErrorString(e));
    }
    return check_launch("alo_conv3x3_nhwc");
}
