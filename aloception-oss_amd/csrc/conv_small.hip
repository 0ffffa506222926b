// 3x3 / stride 1 / padding 1 convolution of channels-last bf16 maps with FEW channels (Cin in {16, 32, 64}, Cout <= 32): the fine
// levels of PanopticHead's mask decoder — lay4 (64 -> 32 at stride 8), lay5 (32 -> 16 at stride 4) and out_lay (16 -> 1) over B*Q maps
// (alonet/detr_panoptic/nn/FPNstyle.py:28-33,76-84; BASELINE configs[4]: 128 maps of up to 200 x 334 pixels).
//
// These layers are memory-bound (lay5: 547 MB in, 274 MB out for 39 GFLOP), and the library kernels reach 0.4-0.9 TB/s on them
// (0.94 / 0.66 / 0.48 ms).  Here a 4-wave workgroup owns an 8 x 16 pixel tile of one map: the 10 x 18 pixel neighbourhood goes to LDS
// once (whole 16-byte pieces, zero outside the map), every wave multiplies its 2 x 16 pixels with v_mfma_f32_32x32x16_bf16 — the
// product transposed (weights = row operand), so a lane ends up with ONE pixel and four consecutive output channels per accumulator
// quad: bias, bf16 rounding and 8-byte stores, no LDS round trip on the way out.  The weights arrive pre-packed in the row operand's
// fragment order (alo_hip.conv3x3_small packs and caches them) and stream through the L1 (<= 36 KB per layer, shared by every tile).
#include "common.hpp"

namespace alo {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16x8_t as_frag(const u32x4& v) {
    union { u32x4 u; bf16x8_t b; } x;
    x.u = v;
    return x.b;
}

constexpr int kTileH = 8, kTileW = 16;             // output pixels per workgroup: 4 waves x (2 rows x 16 pixels)
constexpr int kHaloH = kTileH + 2, kHaloW = kTileW + 2;

struct SmallConvDims {
    int N, H, W, Cout;        // Cout actually stored (1, or a multiple of 4 up to 32)
    int tiles_x, tiles_y;
};

template <int CIN>
__global__ void __launch_bounds__(256)
conv3x3_small_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wfrag, const float* __restrict__ bias32,
                     bf16_t* __restrict__ Y, const SmallConvDims dm) {
    constexpr int PS = CIN * 2 + 16;               // LDS bytes per staged pixel: + 16 keeps the 16-byte fragment reads conflict-free
    constexpr int PIECES = CIN / 8;                // 16-byte pieces per pixel
    __shared__ __attribute__((aligned(16))) unsigned char tile[kHaloH * kHaloW * PS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tiles = dm.tiles_x * dm.tiles_y;
    const int n = blockIdx.x / tiles, t = blockIdx.x - n * tiles;
    const int y0 = (t / dm.tiles_x) * kTileH, x0 = (t % dm.tiles_x) * kTileW;
    const bf16_t* xn = X + (size_t)n * dm.H * dm.W * CIN;

    // ---- the tile's neighbourhood -> LDS ---------------------------------------------------------------------------------------
    for (int i = tid; i < kHaloH * kHaloW * PIECES; i += 256) {
        const int px = i / PIECES, piece = i - px * PIECES;
        const int ry = px / kHaloW, rx = px - ry * kHaloW;
        const int gy = y0 + ry - 1, gx = x0 + rx - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W)
            v = *reinterpret_cast<const u32x4*>(xn + ((size_t)gy * dm.W + gx) * CIN + piece * 8);
        *reinterpret_cast<u32x4*>(tile + px * PS + piece * 16) = v;
    }
    __syncthreads();

    // ---- 32 pixels per wave: column p of the product = pixel (ty, tx) -------------------------------------------------------------
    const int p = lane & 31, kg = lane >> 5;
    const int ty = 2 * wave + (p >> 4), tx = p & 15;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias32[8 * (r >> 2) + 4 * kg + (r & 3)];   // row of register r: 8 (r / 4) + 4 kg + r % 4
    const u32x4* wf = reinterpret_cast<const u32x4*>(Wfrag) + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const unsigned char* src = tile + ((ty + tap / 3) * kHaloW + tx + tap % 3) * PS + kg * 16;
#pragma unroll
        for (int cs = 0; cs < CIN / 16; ++cs) {
            const u32x4 a = wf[(tap * (CIN / 16) + cs) * 64];
            const u32x4 b = *reinterpret_cast<const u32x4*>(src + cs * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(b), acc, 0, 0, 0);
        }
    }
    // ---- store: this lane's pixel, channels 8 j + 4 kg .. + 3 from registers 4 j .. 4 j + 3 ------------------------------------------
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < dm.H && gx < dm.W) {
        bf16_t* yp = Y + (((size_t)n * dm.H + gy) * dm.W + gx) * dm.Cout;
        if (dm.Cout == 1) {
            if (kg == 0) yp[0].bits = (unsigned short)(pack_bf16x2(acc[0], 0.f) & 0xffffu);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 8 * j + 4 * kg;
                if (c < dm.Cout)
                    *reinterpret_cast<u32x2*>(yp + c) = u32x2{pack_bf16x2(acc[4 * j], acc[4 * j + 1]), pack_bf16x2(acc[4 * j + 2], acc[4 * j + 3])};
            }
        }
    }
}

template <int CIN>
int launch_small(const void* x, const void* wfrag, const void* bias32, void* y, SmallConvDims dm, hipStream_t stream) {
    void* args[] = {&x, &wfrag, &bias32, &y, &dm};
    const long blocks = (long)dm.N * dm.tiles_x * dm.tiles_y;
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(conv3x3_small_kernel<CIN>), dim3((unsigned)blocks), dim3(256), args, 0, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_conv3x3_small_nhwc: %s", hipGetErrorString(e));
    return check_launch("alo_conv3x3_small_nhwc");
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" int alo_conv3x3_small_nhwc(const void* x, const void* w_frag, const void* bias32, void* y, int N, int H, int W, int Cin,
                                      int Cout, int dtype, void* stream) {
    ALO_REQUIRE(x && w_frag && bias32 && y, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_small_nhwc: null pointer argument");
    ALO_REQUIRE(N > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_small_nhwc: N, H, W must be positive");
    ALO_REQUIRE(Cin == 16 || Cin == 32 || Cin == 64, ALO_ERR_UNSUPPORTED, "alo_conv3x3_small_nhwc: Cin must be 16, 32 or 64 (got %d)", Cin);
    ALO_REQUIRE(Cout == 1 || (Cout > 0 && Cout <= 32 && Cout % 4 == 0), ALO_ERR_UNSUPPORTED,
                "alo_conv3x3_small_nhwc: Cout must be 1 or a multiple of 4 up to 32 (got %d)", Cout);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_conv3x3_small_nhwc: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)w_frag | (uintptr_t)bias32) & 15) == 0 && ((uintptr_t)y & (Cout == 1 ? 1 : 7)) == 0,
                ALO_ERR_INVALID_ARGUMENT, "alo_conv3x3_small_nhwc: x, w_frag, bias32 must be 16-byte aligned (y: 8-byte, 2 for Cout = 1)");
    SmallConvDims dm;
    dm.N = N; dm.H = H; dm.W = W; dm.Cout = Cout;
    dm.tiles_x = (W + kTileW - 1) / kTileW; dm.tiles_y = (H + kTileH - 1) / kTileH;
    ALO_REQUIRE((long)N * dm.tiles_x * dm.tiles_y < 0x7fffffffL, ALO_ERR_UNSUPPORTED, "alo_conv3x3_small_nhwc: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Cin == 16) return launch_small<16>(x, w_frag, bias32, y, dm, st);
    if (Cin == 32) return launch_small<32>(x, w_frag, bias32, y, dm, st);
    return launch_small<64>(x, w_frag, bias32, y, dm, st);
}
