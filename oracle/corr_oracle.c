/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of RAFT's CorrBlock (all-pairs correlation volume, average-pool pyramid, windowed
 * bilinear lookup).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Semantics followed (paths relative to /root/reference):
 *   volume    alonet/raft/corr.py:52-60    corr[b,i,j] = <fmap1[b,:,i], fmap2[b,:,j]> / sqrt(C)
 *   pyramid   alonet/raft/corr.py:13-27    F.avg_pool2d(k=2, s=2) applied 3x on (B*HW, 1, h, w)  (floor sizes)
 *   lookup    alonet/raft/corr.py:29-50    window axis 0 offsets x, axis 1 offsets y (meshgrid(dy,dx) quirk)
 *             alonet/raft/utils/utils.py:5-19   x -> 2x/(W-1)-1, F.grid_sample(align_corners=True, zeros)
 *   grid_sample(bilinear, align_corners=True, padding zeros) is torch's ATen definition:
 *             ix = ((g+1)/2)*(W-1); corners floor/floor+1; out-of-range corners contribute 0.
 *
 * The volume accumulates in double and rounds once to float: it is the more accurate side of every
 * comparison (torch's fp32 matmul and the fp32 MFMA kernel each differ from it by summation order only).
 * Pinned by tests/golden/g6_corr.npz (generated from the reference's CorrBlock itself).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* fmap1, fmap2: (B,C,HW) float.  corr: (B,HW,HW) float. */
int oracle_corr_volume(const float* f1, const float* f2, float* corr, int B, int C, int HW) {
    const double scale = 1.0 / sqrt((double)C);
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)HW);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < HW; ++i) {
                for (int j = 0; j < HW; ++j) acc[j] = 0.0;
                for (int c = 0; c < C; ++c) {
                    const double a = f1[((long)b * C + c) * HW + i];
                    const float* row = f2 + ((long)b * C + c) * HW;
                    for (int j = 0; j < HW; ++j) acc[j] += a * (double)row[j];
                }
                float* o = corr + ((long)b * HW + i) * HW;
                for (int j = 0; j < HW; ++j) o[j] = (float)(acc[j] * scale);
            }
        }
        free(acc);
    }
    return 0;
}

/* in: (n, h, w) -> out: (n, h/2, w/2), 2x2 mean, trailing odd row/col dropped (avg_pool2d floor mode). */
int oracle_avg_pool2(const float* in, float* out, long n, int h, int w) {
    const int ho = h / 2, wo = w / 2;
#pragma omp parallel for schedule(static)
    for (long k = 0; k < n; ++k) {
        const float* src = in + k * (long)h * w;
        float* dst = out + k * (long)ho * wo;
        for (int y = 0; y < ho; ++y)
            for (int x = 0; x < wo; ++x) {
                const float* p = src + (long)(2 * y) * w + 2 * x;
                /* ATen sums the window then divides by the pool size */
                dst[(long)y * wo + x] = (p[0] + p[1] + p[w] + p[w + 1]) / 4.0f;
            }
    }
    return 0;
}

static inline float tap(const float* img, int h, int w, int y, int x) {
    return (y >= 0 && y < h && x >= 0 && x < w) ? img[(long)y * w + x] : 0.0f;
}

/* pyr[l]: (B*H*W, h_l, w_l) with (h_0,w_0) = (H,W), h_{l+1} = h_l/2.  coords: (B,2,H,W), channel 0 = x.
 * out: (B, num_levels*(2r+1)^2, H, W).  All arithmetic in float, in the reference's order. */
int oracle_corr_lookup(const float* const* pyr, const int32_t* lvl_hw, const float* coords, float* out, int B, int H,
                       int W, int r, int num_levels) {
    const int win = 2 * r + 1;
    const int HW = H * W;
    const int CH = num_levels * win * win;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < HW; ++i) {
            const float cx = coords[((long)b * 2 + 0) * HW + i];
            const float cy = coords[((long)b * 2 + 1) * HW + i];
            for (int l = 0; l < num_levels; ++l) {
                const int h = lvl_hw[2 * l], w = lvl_hw[2 * l + 1];
                const float* img = pyr[l] + ((long)b * HW + i) * (long)h * w;
                const float div = (float)(1 << l);
                const float x0 = cx / div, y0 = cy / div; /* centroid_lvl, corr.py:42 */
                for (int a = 0; a < win; ++a) {           /* first window axis  -> x offset (corr.py:37-44) */
                    for (int c = 0; c < win; ++c) {       /* second window axis -> y offset */
                        const float px = x0 + (float)(a - r);
                        const float py = y0 + (float)(c - r);
                        /* utils.py:8-9 then ATen's unnormalize for align_corners=True */
                        const float gx = 2.0f * px / (float)(w - 1) - 1.0f;
                        const float gy = 2.0f * py / (float)(h - 1) - 1.0f;
                        const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1);
                        const float iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
                        const float fx = floorf(ix), fy = floorf(iy);
                        const int xw = (int)fx, yn = (int)fy;
                        const float tx = ix - fx, ty = iy - fy;
                        const float nw = (1.0f - tx) * (1.0f - ty), ne = tx * (1.0f - ty);
                        const float sw = (1.0f - tx) * ty, se = tx * ty;
                        float v = 0.0f;
                        if (isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f) {
                            v = tap(img, h, w, yn, xw) * nw + tap(img, h, w, yn, xw + 1) * ne +
                                tap(img, h, w, yn + 1, xw) * sw + tap(img, h, w, yn + 1, xw + 1) * se;
                        }
                        const int ch = l * win * win + a * win + c;
                        out[((long)b * CH + ch) * HW + i] = v;
                    }
                }
            }
        }
    }
    return 0;
}
