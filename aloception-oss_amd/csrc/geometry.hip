// Padding-mask pyramid, valid ratios and the encoder's reference points in three small kernels.
//
// Stock, DeformableDETR derives them with ~120 tiny PyTorch kernels per step: one F.interpolate of the (B, 1, H, W) frame mask per
// level (alonet/detr/backbone.py:127-128 bilinear + .to(bool); deformable_detr.py:147 nearest for the extra level), flatten + cat
// (deformable_transformer.py:334-338), get_valid_ratio per level (:318-323: sums over the first column / first row) and
// get_reference_points (:136-149: arange, meshgrid, divide, stack per level).  All of it is arithmetic on a few hundred thousand
// elements; the time was launch latency.
#include "common.hpp"

namespace alo {
namespace {

constexpr int kMaxGeoLevels = 8;

struct GeoDims {
    int B, H, W, L, S;
    int h[kMaxGeoLevels], w[kMaxGeoLevels], start[kMaxGeoLevels];
    unsigned nearest;   // bit l set: level l is resized with mode="nearest", else bilinear (align_corners = False)
};

__device__ __forceinline__ int level_of(const GeoDims& dm, int s) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxGeoLevels; ++i)
        if (i < dm.L && s >= dm.start[i]) l = i;
    return l;
}

// out[b][s] = F.interpolate(mask.float(), (h_l, w_l), mode)[b, 0, y, x] != 0, in ATen's own arithmetic (UpSample.cuh:
// area_pixel_compute_scale / _source_index, nearest_neighbor_compute_source_index)
template <typename MT>
__global__ void __launch_bounds__(256)
mask_pyramid_kernel(const MT* __restrict__ mask, unsigned char* __restrict__ out, const GeoDims dm) {
    const long total = (long)dm.B * dm.S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / dm.S), s = (int)(i - (long)b * dm.S);
        const int l = level_of(dm, s);
        const int hl = dm.h[l], wl = dm.w[l];
        const int p = s - dm.start[l], y = p / wl, x = p - y * wl;
        const MT* m = mask + (size_t)b * dm.H * dm.W;
        const float sh = (float)dm.H / (float)hl, sw = (float)dm.W / (float)wl;
        bool on;
        if ((dm.nearest >> l) & 1u) {
            const int iy = min((int)floorf((float)y * sh), dm.H - 1), ix = min((int)floorf((float)x * sw), dm.W - 1);
            on = m[(size_t)iy * dm.W + ix] != (MT)0;
        } else {
            float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            fx = fx < 0.f ? 0.f : fx;
            const int y1 = (int)fy, x1 = (int)fx;
            const int yp = y1 < dm.H - 1 ? 1 : 0, xp = x1 < dm.W - 1 ? 1 : 0;
            const float ly = fy - (float)y1, lx = fx - (float)x1;
            const float m00 = m[(size_t)y1 * dm.W + x1] != (MT)0 ? 1.f : 0.f, m01 = m[(size_t)y1 * dm.W + x1 + xp] != (MT)0 ? 1.f : 0.f;
            const float m10 = m[(size_t)(y1 + yp) * dm.W + x1] != (MT)0 ? 1.f : 0.f, m11 = m[(size_t)(y1 + yp) * dm.W + x1 + xp] != (MT)0 ? 1.f : 0.f;
            // every term is >= 0: the interpolated value is non-zero iff one of them is
            on = (1.f - ly) * ((1.f - lx) * m00 + lx * m01) + ly * ((1.f - lx) * m10 + lx * m11) != 0.f;
        }
        out[i] = on ? 1 : 0;
    }
}

// valid_ratios[b][l] = (count of un-padded pixels in the first ROW / w_l, in the first COLUMN / h_l); one wave per (b, l)
__global__ void __launch_bounds__(64)
valid_ratio_kernel(const unsigned char* __restrict__ mflat, float* __restrict__ ratios, const GeoDims dm) {
    const int b = blockIdx.x / dm.L, l = blockIdx.x % dm.L, lane = threadIdx.x;
    const unsigned char* m = mflat + (size_t)b * dm.S + dm.start[l];
    const int hl = dm.h[l], wl = dm.w[l];
    int vw = 0, vh = 0;
    for (int x = lane; x < wl; x += 64) vw += m[x] ? 0 : 1;
    for (int y = lane; y < hl; y += 64) vh += m[(size_t)y * wl] ? 0 : 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vw += __shfl_xor(vw, o, 64); vh += __shfl_xor(vh, o, 64); }
    if (lane == 0) {
        // `tensor / python_int` in PyTorch multiplies by the reciprocal of the scalar (BinaryDivTrueKernel.cu): same here, bit for bit
        ratios[((size_t)b * dm.L + l) * 2] = (float)vw * (1.0f / (float)wl);
        ratios[((size_t)b * dm.L + l) * 2 + 1] = (float)vh * (1.0f / (float)hl);
    }
}

// ref[b][s][l'][0:2] = ((x + 0.5) / (vr[b][l][0] * w_l) * vr[b][l'][0], (y + 0.5) / (vr[b][l][1] * h_l) * vr[b][l'][1]), s = pixel (y, x) of level l
__global__ void __launch_bounds__(256)
encoder_reference_points_kernel(const float* __restrict__ ratios, float* __restrict__ ref, const GeoDims dm) {
    const long total = (long)dm.B * dm.S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / dm.S), s = (int)(i - (long)b * dm.S);
        const int l = level_of(dm, s);
        const int wl = dm.w[l], p = s - dm.start[l], y = p / wl, x = p - y * wl;
        const float* vr = ratios + (size_t)b * dm.L * 2;
        const float rx = ((float)x + 0.5f) / (vr[2 * l] * (float)wl), ry = ((float)y + 0.5f) / (vr[2 * l + 1] * (float)dm.h[l]);
        float* o = ref + (size_t)i * dm.L * 2;
        for (int k = 0; k < dm.L; ++k) {
            o[2 * k] = rx * vr[2 * k];
            o[2 * k + 1] = ry * vr[2 * k + 1];
        }
    }
}

int fill_dims(GeoDims& dm, const char* what, int B, int H, int W, int L, const int* shapes_host, unsigned nearest) {
    ALO_REQUIRE(B > 0 && L > 0 && L <= kMaxGeoLevels && shapes_host, ALO_ERR_INVALID_ARGUMENT, "%s: B, L (<= %d) and the level shapes are required", what, kMaxGeoLevels);
    dm.B = B; dm.H = H; dm.W = W; dm.L = L; dm.nearest = nearest;
    int s = 0;
    for (int l = 0; l < kMaxGeoLevels; ++l) {
        dm.h[l] = l < L ? shapes_host[2 * l] : 1;
        dm.w[l] = l < L ? shapes_host[2 * l + 1] : 1;
        dm.start[l] = s;
        if (l < L) {
            ALO_REQUIRE(dm.h[l] > 0 && dm.w[l] > 0, ALO_ERR_INVALID_ARGUMENT, "%s: level %d has an empty shape", what, l);
            s += dm.h[l] * dm.w[l];
        }
    }
    dm.S = s;
    return ALO_OK;
}

unsigned geo_blocks(long n) {
    long blocks = (n + 255) / 256;
    return (unsigned)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks));
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" int alo_mask_pyramid(const void* frame_mask, int mask_is_float, unsigned char* mask_flat, float* valid_ratios, int B,
                                int H, int W, int L, const int* level_shapes_host, unsigned nearest_levels, void* stream) {
    ALO_REQUIRE(frame_mask && mask_flat && valid_ratios, ALO_ERR_INVALID_ARGUMENT, "alo_mask_pyramid: null pointer argument");
    ALO_REQUIRE(H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_mask_pyramid: empty frame");
    GeoDims dm;
    if (int rc = fill_dims(dm, "alo_mask_pyramid", B, H, W, L, level_shapes_host, nearest_levels)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* args[] = {&frame_mask, &mask_flat, &dm};
    const void* kern = mask_is_float ? reinterpret_cast<const void*>(mask_pyramid_kernel<float>)
                                     : reinterpret_cast<const void*>(mask_pyramid_kernel<unsigned char>);
    hipError_t e = hipLaunchKernel(kern, dim3(geo_blocks((long)B * dm.S)), dim3(256), args, 0, s);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_mask_pyramid: %s", hipGetErrorString(e));
    void* args2[] = {&mask_flat, &valid_ratios, &dm};
    e = hipLaunchKernel(reinterpret_cast<const void*>(valid_ratio_kernel), dim3((unsigned)(B * L)), dim3(64), args2, 0, s);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_mask_pyramid: %s", hipGetErrorString(e));
    return check_launch("alo_mask_pyramid");
}

extern "C" int alo_encoder_reference_points(const float* valid_ratios, float* reference_points, int B, int L,
                                            const int* level_shapes_host, void* stream) {
    ALO_REQUIRE(valid_ratios && reference_points, ALO_ERR_INVALID_ARGUMENT, "alo_encoder_reference_points: null pointer argument");
    GeoDims dm;
    if (int rc = fill_dims(dm, "alo_encoder_reference_points", B, 1, 1, L, level_shapes_host, 0u)) return rc;
    void* args[] = {&valid_ratios, &reference_points, &dm};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(encoder_reference_points_kernel), dim3(geo_blocks((long)B * dm.S)),
                                   dim3(256), args, 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_encoder_reference_points: %s", hipGetErrorString(e));
    return check_launch("alo_encoder_reference_points");
}

// ---- panoptic post-processing: logits at mask resolution -> one-hot instance masks at frame resolution ----------------------------
// alonet/detr_panoptic/detr_panoptic.py:96-110 (inference): F.interpolate(pred_masks.float(), frame_size, "bilinear") -> sigmoid ->
// F.threshold(., maskth, 0) -> per pixel the arg-max query gets 1 unless no query passed the threshold.  Stock: ~10 element-wise
// passes over (B, Q, H, W) fp32 / int64 tensors (1.1 GB each at 8 x 16 x 800 x 1333).  Here: one pass, the only full-size tensor
// touched is the int64 output.
namespace alo {
namespace {
struct OnehotDims {
    int B, Q, h, w, H, W;
    float thr;
};

__global__ void __launch_bounds__(256)
panoptic_onehot_kernel(const float* __restrict__ logits, long long* __restrict__ out, const OnehotDims dm) {
    const long total = (long)dm.B * dm.H * dm.W;
    const float sh = (float)dm.h / (float)dm.H, sw = (float)dm.w / (float)dm.W;   // ATen: area_pixel_compute_scale, align_corners = False
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / ((long)dm.H * dm.W));
        const int rem = (int)(i - (long)b * dm.H * dm.W), y = rem / dm.W, x = rem - y * dm.W;
        float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y1 = (int)fy, x1 = (int)fx;
        const int yp = y1 < dm.h - 1 ? 1 : 0, xp = x1 < dm.w - 1 ? 1 : 0;
        const float ly1 = fy - (float)y1, lx1 = fx - (float)x1, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* base = logits + (size_t)b * dm.Q * dm.h * dm.w + (size_t)y1 * dm.w + x1;
        float best = 0.f;
        int arg = -1;
        for (int q = 0; q < dm.Q; ++q) {
            const float* p = base + (size_t)q * dm.h * dm.w;
            const float v = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[yp * dm.w] + lx1 * p[yp * dm.w + xp]);
            float s = 1.f / (1.f + expf(-v));
            s = s > dm.thr ? s : 0.f;          // F.threshold(s, thr, 0)
            if (s > best) { best = s; arg = q; }   // strict: ties keep the lowest query, as torch.argmax
        }
        long long* o = out + (size_t)b * dm.Q * dm.H * dm.W + rem;
        for (int q = 0; q < dm.Q; ++q) o[(size_t)q * dm.H * dm.W] = q == arg ? 1 : 0;
    }
}
}  // namespace
}  // namespace alo

extern "C" int alo_panoptic_onehot(const float* mask_logits, long long* onehot, int B, int Q, int h, int w, int H, int W,
                                   float threshold, void* stream) {
    ALO_REQUIRE(mask_logits && onehot, ALO_ERR_INVALID_ARGUMENT, "alo_panoptic_onehot: null pointer argument");
    ALO_REQUIRE(B > 0 && Q > 0 && h > 0 && w > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT, "alo_panoptic_onehot: sizes must be positive");
    ALO_REQUIRE(threshold >= 0.f, ALO_ERR_UNSUPPORTED, "alo_panoptic_onehot: the threshold must be non-negative (got %f)", (double)threshold);
    alo::OnehotDims dm;
    dm.B = B; dm.Q = Q; dm.h = h; dm.w = w; dm.H = H; dm.W = W; dm.thr = threshold;
    void* args[] = {&mask_logits, &onehot, &dm};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(alo::panoptic_onehot_kernel), dim3(alo::geo_blocks((long)B * H * W) * 4), dim3(256),
                                   args, 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_panoptic_onehot: %s", hipGetErrorString(e));
    return check_launch("alo_panoptic_onehot");
}
