// GroupNorm over channels-last rows, written straight into the encoder's flattened (B, S, C) source buffer:
// `input_proj[l][1]` (nn.GroupNorm(32, hidden)) + the `src.flatten(2).transpose(1, 2)` / torch.cat of DeformableTransformer.forward
// (alonet/deformable_detr/deformable_detr.py:75-76,141-151, deformable_transformer.py:331-337).
//
// Stock, ATen's GroupNorm wants NCHW: the NHWC projection output is transposed (a full copy), normalised, and copied twice more on
// its way into the flattened layout.  Here: one statistics pass (per 256-row chunk a Welford triple per group, combined with Chan's
// formula — deterministic, no atomics) and one normalise pass whose output rows land at the level's offset of the flat buffer.
#include "common.hpp"

namespace alo {
namespace {

constexpr int kGnThreads = 256;
constexpr int kGnRows = 256;   // rows per workgroup, both passes

__device__ __forceinline__ void unpack8(const u32x4& x, float (&v)[8]) {
    v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xffff0000u);
    v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xffff0000u);
    v[4] = __uint_as_float(x.z << 16); v[5] = __uint_as_float(x.z & 0xffff0000u);
    v[6] = __uint_as_float(x.w << 16); v[7] = __uint_as_float(x.w & 0xffff0000u);
}

// Chan et al.: merge (n_b, mean_b, M2_b) into (n, mean, M2)
__device__ __forceinline__ void welford_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
}

struct GnDims {
    int HW, C, groups, nchunks;
    long y_batch_stride;   // elements between the batches of the output
    float eps;
    int relu;              // alo_groupnorm_rows_act: y = max(y, 0)
};

// workspace[b][chunk][group] = (n, mean, M2) of the chunk's rows
__global__ void __launch_bounds__(kGnThreads)
groupnorm_stats_kernel(const bf16_t* __restrict__ X, float* __restrict__ ws, const GnDims dm) {
    __shared__ float part[kGnThreads][3];
    const int tid = threadIdx.x, tpr = dm.C / 8, rpp = kGnThreads / tpr;   // threads per row, rows per pass
    const int cg = tid % tpr, r0 = tid / tpr;
    const int b = blockIdx.y, row_begin = blockIdx.x * kGnRows;
    const int row_end = row_begin + kGnRows < dm.HW ? row_begin + kGnRows : dm.HW;
    const bf16_t* xb = X + ((size_t)b * dm.HW) * dm.C + cg * 8;
    float s = 0.f, ss = 0.f, n = 0.f;
    for (int r = row_begin + r0; r < row_end; r += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + (size_t)r * dm.C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += v[i]; ss += v[i] * v[i]; }
        n += 8.f;
    }
    const float mean = n > 0.f ? s / n : 0.f;
    part[tid][0] = n; part[tid][1] = mean; part[tid][2] = n > 0.f ? fmaxf(ss - s * mean, 0.f) : 0.f;
    __syncthreads();
    if (tid < dm.groups) {   // group g = channel slices g * (gs / 8) .. of every row slot
        const int per = tpr / dm.groups;
        float gn = 0.f, gm = 0.f, g2 = 0.f;
        for (int rr = 0; rr < rpp; ++rr)
            for (int j = 0; j < per; ++j) {
                const int t = rr * tpr + tid * per + j;
                welford_merge(gn, gm, g2, part[t][0], part[t][1], part[t][2]);
            }
        float* o = ws + (((size_t)b * dm.nchunks + blockIdx.x) * dm.groups + tid) * 3;
        o[0] = gn; o[1] = gm; o[2] = g2;
    }
}

__global__ void __launch_bounds__(kGnThreads)
groupnorm_apply_kernel(const bf16_t* __restrict__ X, const float* __restrict__ ws, const bf16_t* __restrict__ gamma,
                       const bf16_t* __restrict__ beta, bf16_t* __restrict__ Y, const GnDims dm) {
    __shared__ float part[kGnThreads][3];
    __shared__ float stat[kGnThreads][2];   // mean, rstd per group
    const int tid = threadIdx.x, b = blockIdx.y;
    {   // every workgroup re-derives the batch's group statistics from the chunk triples (a few KB out of L2)
        const int g = tid % dm.groups, stripe = tid / dm.groups, nstripes = kGnThreads / dm.groups;
        float gn = 0.f, gm = 0.f, g2 = 0.f;
        for (int c = stripe; c < dm.nchunks; c += nstripes) {
            const float* p = ws + (((size_t)b * dm.nchunks + c) * dm.groups + g) * 3;
            welford_merge(gn, gm, g2, p[0], p[1], p[2]);
        }
        part[tid][0] = gn; part[tid][1] = gm; part[tid][2] = g2;
        __syncthreads();
        if (tid < dm.groups) {
            gn = 0.f; gm = 0.f; g2 = 0.f;
            for (int st = 0; st < nstripes; ++st) {
                const int t = st * dm.groups + tid;
                welford_merge(gn, gm, g2, part[t][0], part[t][1], part[t][2]);
            }
            stat[tid][0] = gm;
            stat[tid][1] = rsqrtf(g2 / gn + dm.eps);
        }
        __syncthreads();
    }
    const int tpr = dm.C / 8, rpp = kGnThreads / tpr, cg = tid % tpr, r0 = tid / tpr;
    const int g = cg / (tpr / dm.groups);
    const float mean = stat[g][0], rstd = stat[g][1];
    float ga[8], be[8];
    unpack8(*reinterpret_cast<const u32x4*>(gamma + cg * 8), ga);
    unpack8(*reinterpret_cast<const u32x4*>(beta + cg * 8), be);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ga[i] *= rstd; be[i] -= mean * ga[i]; }   // y = x * ga + be
    const bool relu = dm.relu != 0;   // alo_groupnorm_rows_act: y = relu(y), NaN kept as torch's relu keeps it
    const int row_begin = blockIdx.x * kGnRows;
    const int row_end = row_begin + kGnRows < dm.HW ? row_begin + kGnRows : dm.HW;
    const bf16_t* xb = X + ((size_t)b * dm.HW) * dm.C + cg * 8;
    bf16_t* yb = Y + (size_t)b * dm.y_batch_stride + cg * 8;
    for (int r = row_begin + r0; r < row_end; r += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + (size_t)r * dm.C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = fmaf(v[i], ga[i], be[i]);
            if (relu) v[i] = v[i] < 0.f ? 0.f : v[i];
        }
        u32x4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<u32x4*>(yb + (size_t)r * dm.C) = o;
    }
}


// ---- groups narrower than one 8-channel slice (PanopticHead's mask decoder: GroupNorm(8, 32) and GroupNorm(8, 16) over B*Q maps of
// 100 x 167 / 200 x 334 pixels, alonet/detr_panoptic/nn/FPNstyle.py:24-36).  CPG = channels per group in {2, 4}: a thread's 8 channels
// span NSUB = 8 / CPG groups, so it keeps NSUB running sums; everything else is the scheme above (chunk triples, Chan's merge).
template <int CPG>
__global__ void __launch_bounds__(kGnThreads)
groupnorm_stats_small_kernel(const bf16_t* __restrict__ X, float* __restrict__ ws, const GnDims dm) {
    constexpr int NSUB = 8 / CPG;
    __shared__ float part[kGnThreads][NSUB][3];
    const int tid = threadIdx.x, tpr = dm.C / 8, rpp = kGnThreads / tpr;
    const int cg = tid % tpr, r0 = tid / tpr;
    const int b = blockIdx.y, row_begin = blockIdx.x * kGnRows;
    const int row_end = row_begin + kGnRows < dm.HW ? row_begin + kGnRows : dm.HW;
    const bf16_t* xb = X + ((size_t)b * dm.HW) * dm.C + cg * 8;
    float s[NSUB], ss[NSUB], n = 0.f;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) { s[j] = 0.f; ss[j] = 0.f; }
    for (int r = row_begin + r0; r < row_end; r += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + (size_t)r * dm.C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i / CPG] += v[i]; ss[i / CPG] += v[i] * v[i]; }
        n += (float)CPG;
    }
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const float mean = n > 0.f ? s[j] / n : 0.f;
        part[tid][j][0] = n; part[tid][j][1] = mean; part[tid][j][2] = n > 0.f ? fmaxf(ss[j] - s[j] * mean, 0.f) : 0.f;
    }
    __syncthreads();
    if (tid < dm.groups) {   // group g = sub-group g % NSUB of slice g / NSUB, over every row slot
        const int slice = tid / NSUB, sub = tid % NSUB;
        float gn = 0.f, gm = 0.f, g2 = 0.f;
        for (int rr = 0; rr < rpp; ++rr) {
            const int t = rr * tpr + slice;
            welford_merge(gn, gm, g2, part[t][sub][0], part[t][sub][1], part[t][sub][2]);
        }
        float* o = ws + (((size_t)b * dm.nchunks + blockIdx.x) * dm.groups + tid) * 3;
        o[0] = gn; o[1] = gm; o[2] = g2;
    }
}

template <int CPG>
__global__ void __launch_bounds__(kGnThreads)
groupnorm_apply_small_kernel(const bf16_t* __restrict__ X, const float* __restrict__ ws, const bf16_t* __restrict__ gamma,
                             const bf16_t* __restrict__ beta, bf16_t* __restrict__ Y, const GnDims dm) {
    constexpr int NSUB = 8 / CPG;
    __shared__ float part[kGnThreads][3];
    __shared__ float stat[kGnThreads][2];
    const int tid = threadIdx.x, b = blockIdx.y;
    {
        const int g = tid % dm.groups, stripe = tid / dm.groups, nstripes = kGnThreads / dm.groups;
        float gn = 0.f, gm = 0.f, g2 = 0.f;
        for (int c = stripe; c < dm.nchunks; c += nstripes) {
            const float* p = ws + (((size_t)b * dm.nchunks + c) * dm.groups + g) * 3;
            welford_merge(gn, gm, g2, p[0], p[1], p[2]);
        }
        part[tid][0] = gn; part[tid][1] = gm; part[tid][2] = g2;
        __syncthreads();
        if (tid < dm.groups) {
            gn = 0.f; gm = 0.f; g2 = 0.f;
            for (int st = 0; st < nstripes; ++st) {
                const int t = st * dm.groups + tid;
                welford_merge(gn, gm, g2, part[t][0], part[t][1], part[t][2]);
            }
            stat[tid][0] = gm;
            stat[tid][1] = rsqrtf(g2 / gn + dm.eps);
        }
        __syncthreads();
    }
    const int tpr = dm.C / 8, rpp = kGnThreads / tpr, cg = tid % tpr, r0 = tid / tpr;
    float ga[8], be[8];
    unpack8(*reinterpret_cast<const u32x4*>(gamma + cg * 8), ga);
    unpack8(*reinterpret_cast<const u32x4*>(beta + cg * 8), be);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = cg * NSUB + i / CPG;
        ga[i] *= stat[g][1];
        be[i] -= stat[g][0] * ga[i];
    }
    const bool relu = dm.relu != 0;
    const int row_begin = blockIdx.x * kGnRows;
    const int row_end = row_begin + kGnRows < dm.HW ? row_begin + kGnRows : dm.HW;
    const bf16_t* xb = X + ((size_t)b * dm.HW) * dm.C + cg * 8;
    bf16_t* yb = Y + (size_t)b * dm.y_batch_stride + cg * 8;
    for (int r = row_begin + r0; r < row_end; r += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + (size_t)r * dm.C), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = fmaf(v[i], ga[i], be[i]);
            if (relu) v[i] = v[i] < 0.f ? 0.f : v[i];   // NaN kept, as torch's relu keeps it
        }
        u32x4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<u32x4*>(yb + (size_t)r * dm.C) = o;
    }
}

// ---- nearest up-sampling + broadcast add of the mask decoder's FPN steps (FPNstyle.py:60-84):
//   out[bq, y, x, :] = fpn[bq / Q, y, x, :] + x_low[bq, floor(y * h / H), floor(x * w / W), :]        (channels-last, bf16)
// i.e. `_expand(adapter(fpn), Q) + F.interpolate(x, size=(H, W), mode="nearest")` without the repeated copy of the adapter output,
// the up-sampled copy and the add pass (three passes over the stage's largest tensor become one write).  Index arithmetic as ATen's
// nearest kernel: scale = float(in) / out, src = min(int(floorf(dst * scale)), in - 1); the add is fp32, rounded once.
struct UpAddDims {
    int Q, C8, h, w, H, W;
    long total;   // B*Q * H * W * C8
    float sy, sx;
};

__global__ void __launch_bounds__(256)
upsample_add_nhwc_kernel(const bf16_t* __restrict__ xlow, const bf16_t* __restrict__ fpn, bf16_t* __restrict__ out, const UpAddDims dm) {
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < dm.total; idx += (long)gridDim.x * 256L) {
        const int c8 = (int)(idx % dm.C8);
        const long pix = idx / dm.C8;
        const int x = (int)(pix % dm.W);
        const long row = pix / dm.W;
        const int y = (int)(row % dm.H);
        const long bq = row / dm.H, b = bq / dm.Q;
        const int ys = min((int)floorf((float)y * dm.sy), dm.h - 1), xs = min((int)floorf((float)x * dm.sx), dm.w - 1);
        float a[8], f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xlow + (((bq * dm.h + ys) * dm.w + xs) * dm.C8 + c8) * 8), a);
        unpack8(*reinterpret_cast<const u32x4*>(fpn + (((b * dm.H + y) * dm.W + x) * dm.C8 + c8) * 8), f);
        u32x4 o;
        o.x = pack_bf16x2(f[0] + a[0], f[1] + a[1]); o.y = pack_bf16x2(f[2] + a[2], f[3] + a[3]);
        o.z = pack_bf16x2(f[4] + a[4], f[5] + a[5]); o.w = pack_bf16x2(f[6] + a[6], f[7] + a[7]);
        *reinterpret_cast<u32x4*>(out + idx * 8) = o;
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" size_t alo_groupnorm_rows_workspace_bytes(int B, int HW, int groups) {
    if (B <= 0 || HW <= 0 || groups <= 0) return 0;
    return (size_t)B * ((HW + kGnRows - 1) / kGnRows) * groups * 3 * sizeof(float);
}

namespace {
int groupnorm_rows_impl(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW, int C, int groups,
                        float eps, long y_batch_stride, int relu, int dtype, void* stream) {
    ALO_REQUIRE(x && weight && bias && y && workspace, ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: null pointer argument");
    ALO_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0, ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: sizes must be positive");
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_groupnorm_rows: bf16 only (dtype %d)", dtype);
    const int cpg = C % groups == 0 ? C / groups : 0;
    const bool wide = cpg > 0 && cpg % 8 == 0, narrow = cpg == 2 || cpg == 4;
    ALO_REQUIRE(C % 8 == 0 && kGnThreads % (C / 8) == 0 && (wide || narrow) && kGnThreads % groups == 0, ALO_ERR_UNSUPPORTED,
                "alo_groupnorm_rows: needs C / 8 and groups to divide 256 and 2, 4 or a multiple of 8 channels per group (C=%d groups=%d)",
                C, groups);
    ALO_REQUIRE(y_batch_stride >= (long)HW * C && y_batch_stride % 8 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_groupnorm_rows: y_batch_stride must cover HW * C and keep rows 16-byte aligned");
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0,
                ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: pointers must be 16-byte aligned");
    GnDims dm;
    dm.HW = HW; dm.C = C; dm.groups = groups; dm.nchunks = (HW + kGnRows - 1) / kGnRows;
    dm.y_batch_stride = y_batch_stride; dm.eps = eps; dm.relu = relu ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const void* stats = wide ? reinterpret_cast<const void*>(groupnorm_stats_kernel)
                             : (cpg == 4 ? reinterpret_cast<const void*>(groupnorm_stats_small_kernel<4>)
                                         : reinterpret_cast<const void*>(groupnorm_stats_small_kernel<2>));
    const void* apply = wide ? reinterpret_cast<const void*>(groupnorm_apply_kernel)
                             : (cpg == 4 ? reinterpret_cast<const void*>(groupnorm_apply_small_kernel<4>)
                                         : reinterpret_cast<const void*>(groupnorm_apply_small_kernel<2>));
    {
        void* args[] = {&x, &workspace, &dm};
        hipError_t e = hipLaunchKernel(stats, dim3(dm.nchunks, B), dim3(kGnThreads), args, 0, s);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_groupnorm_rows: %s", hipGetErrorString(e));
    }
    {
        void* args[] = {&x, &workspace, &weight, &bias, &y, &dm};
        hipError_t e = hipLaunchKernel(apply, dim3(dm.nchunks, B), dim3(kGnThreads), args, 0, s);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_groupnorm_rows: %s", hipGetErrorString(e));
    }
    return check_launch("alo_groupnorm_rows");
}
}  // namespace

extern "C" int alo_groupnorm_rows(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW,
                                  int C, int groups, float eps, long y_batch_stride, int dtype, void* stream) {
    return groupnorm_rows_impl(x, weight, bias, y, workspace, B, HW, C, groups, eps, y_batch_stride, 0, dtype, stream);
}

extern "C" int alo_groupnorm_rows_act(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW,
                                      int C, int groups, float eps, long y_batch_stride, int relu, int dtype, void* stream) {
    return groupnorm_rows_impl(x, weight, bias, y, workspace, B, HW, C, groups, eps, y_batch_stride, relu, dtype, stream);
}

extern "C" int alo_upsample_add_nhwc(const void* x_low, const void* fpn, void* out, int BQ, int Q, int C, int h, int w, int H, int W,
                                     int dtype, void* stream) {
    ALO_REQUIRE(x_low && fpn && out, ALO_ERR_INVALID_ARGUMENT, "alo_upsample_add_nhwc: null pointer argument");
    ALO_REQUIRE(BQ > 0 && Q > 0 && BQ % Q == 0 && C > 0 && C % 8 == 0 && h > 0 && w > 0 && H > 0 && W > 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_upsample_add_nhwc: BQ must be a positive multiple of Q and C a positive multiple of 8 (BQ=%d Q=%d C=%d)", BQ, Q, C);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_upsample_add_nhwc: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)x_low | (uintptr_t)fpn | (uintptr_t)out) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_upsample_add_nhwc: pointers must be 16-byte aligned");
    UpAddDims dm;
    dm.Q = Q; dm.C8 = C / 8; dm.h = h; dm.w = w; dm.H = H; dm.W = W;
    dm.total = (long)BQ * H * W * dm.C8;
    dm.sy = (float)h / (float)H; dm.sx = (float)w / (float)W;
    long blocks = (dm.total + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    void* args[] = {&x_low, &fpn, &out, &dm};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(upsample_add_nhwc_kernel), dim3((unsigned)blocks), dim3(256), args, 0,
                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_upsample_add_nhwc: %s", hipGetErrorString(e));
    return check_launch("alo_upsample_add_nhwc");
}
