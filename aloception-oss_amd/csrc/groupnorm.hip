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
    const int row_begin = blockIdx.x * kGnRows;
    const int row_end = row_begin + kGnRows < dm.HW ? row_begin + kGnRows : dm.HW;
    const bf16_t* xb = X + ((size_t)b * dm.HW) * dm.C + cg * 8;
    bf16_t* yb = Y + (size_t)b * dm.y_batch_stride + cg * 8;
    for (int r = row_begin + r0; r < row_end; r += rpp) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + (size_t)r * dm.C), v);
        u32x4 o;
        o.x = pack_bf16x2(fmaf(v[0], ga[0], be[0]), fmaf(v[1], ga[1], be[1]));
        o.y = pack_bf16x2(fmaf(v[2], ga[2], be[2]), fmaf(v[3], ga[3], be[3]));
        o.z = pack_bf16x2(fmaf(v[4], ga[4], be[4]), fmaf(v[5], ga[5], be[5]));
        o.w = pack_bf16x2(fmaf(v[6], ga[6], be[6]), fmaf(v[7], ga[7], be[7]));
        *reinterpret_cast<u32x4*>(yb + (size_t)r * dm.C) = o;
    }
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" size_t alo_groupnorm_rows_workspace_bytes(int B, int HW, int groups) {
    if (B <= 0 || HW <= 0 || groups <= 0) return 0;
    return (size_t)B * ((HW + kGnRows - 1) / kGnRows) * groups * 3 * sizeof(float);
}

extern "C" int alo_groupnorm_rows(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW,
                                  int C, int groups, float eps, long y_batch_stride, int dtype, void* stream) {
    ALO_REQUIRE(x && weight && bias && y && workspace, ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: null pointer argument");
    ALO_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0, ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: sizes must be positive");
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_groupnorm_rows: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE(C % 8 == 0 && kGnThreads % (C / 8) == 0 && C % groups == 0 && (C / groups) % 8 == 0 && kGnThreads % groups == 0,
                ALO_ERR_UNSUPPORTED,
                "alo_groupnorm_rows: needs C / 8 and groups to divide 256 and whole 8-channel slices per group (C=%d groups=%d)", C, groups);
    ALO_REQUIRE(y_batch_stride >= (long)HW * C && y_batch_stride % 8 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_groupnorm_rows: y_batch_stride must cover HW * C and keep rows 16-byte aligned");
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0,
                ALO_ERR_INVALID_ARGUMENT, "alo_groupnorm_rows: pointers must be 16-byte aligned");
    GnDims dm;
    dm.HW = HW; dm.C = C; dm.groups = groups; dm.nchunks = (HW + kGnRows - 1) / kGnRows;
    dm.y_batch_stride = y_batch_stride; dm.eps = eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        void* args[] = {&x, &workspace, &dm};
        hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(groupnorm_stats_kernel), dim3(dm.nchunks, B), dim3(kGnThreads),
                                       args, 0, s);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_groupnorm_rows: %s", hipGetErrorString(e));
    }
    {
        void* args[] = {&x, &workspace, &weight, &bias, &y, &dm};
        hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(groupnorm_apply_kernel), dim3(dm.nchunks, B), dim3(kGnThreads),
                                       args, 0, s);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_groupnorm_rows: %s", hipGetErrorString(e));
    }
    return check_launch("alo_groupnorm_rows");
}
