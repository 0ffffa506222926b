/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of multi-scale deformable attention (forward + backward) as the reference's
 * native op defines it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (aloception-oss_amd/) never does.
 *
 * Semantics followed (all paths relative to /root/reference):
 *   forward   alonet/deformable_detr/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299  (per-output loop)
 *             bilinear helper                                        ...cuh:33-84
 *   backward  ...cuh:87-159 (per-sample gradient rule), 301-403 (sum over channels for d loc / d attn)
 *   host      alonet/deformable_detr/ops/src/cuda/ms_deform_attn_cuda.cu:20-153 (layouts, zero-init)
 *
 * Layouts:  value (N,S,M,D)   shapes (L,2) int32 [H,W]   level_start (L) int32
 *           loc (N,Lq,M,L,P,2) last dim (x,y) normalised     attn (N,Lq,M,L,P)
 *           out / grad_out (N,Lq,M*D)
 * Pinned by tests/golden/g1,g2,g3,g8 (generated from the reference's own pure-torch oracle
 * ms_deform_attn_core_pytorch + autograd; tests/golden/make_golden.py).
 *
 * Built twice from this one file: -DREAL=double -DSUF=f64 and -DREAL=float -DSUF=f32.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef REAL
#define REAL double
#define SUF f64
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* One sampling point: image coordinates, validity test and the four corner taps.
 * cuh:285-291 for the mapping/test, cuh:38-78 for the per-corner bounds checks. */
typedef struct {
    int valid;
    int h_low, w_low;
    REAL lh, lw, hh, hw;
} tap_t;

static inline tap_t make_tap(REAL loc_w, REAL loc_h, int H, int W) {
    tap_t t;
    const REAL h_im = loc_h * H - (REAL)0.5;
    const REAL w_im = loc_w * W - (REAL)0.5;
    t.valid = (h_im > -1 && w_im > -1 && h_im < H && w_im < W);
    t.h_low = (int)floor((double)h_im);
    t.w_low = (int)floor((double)w_im);
    t.lh = h_im - t.h_low;
    t.lw = w_im - t.w_low;
    t.hh = 1 - t.lh;
    t.hw = 1 - t.lw;
    return t;
}

int FN(oracle_msda_forward)(const REAL* value, const int32_t* shapes, const int32_t* level_start,
                            const REAL* loc, const REAL* attn, REAL* out, int N, int S, int M, int D, int L,
                            int Lq, int P) {
    const long row = (long)M * D; /* elements per spatial position */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < N; ++b) {
        for (int q = 0; q < Lq; ++q) {
            for (int m = 0; m < M; ++m) {
                REAL* o = out + (((long)b * Lq + q) * M + m) * D;
                for (int c = 0; c < D; ++c) o[c] = 0;
                const long sidx = ((long)b * Lq + q) * M + m; /* "sampling_index", cuh:256 */
                const REAL* lp = loc + sidx * L * P * 2;
                const REAL* ap = attn + sidx * L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = shapes[2 * l], W = shapes[2 * l + 1];
                    const REAL* v = value + ((long)b * S + level_start[l]) * row + (long)m * D;
                    for (int p = 0; p < P; ++p, lp += 2, ++ap) {
                        const tap_t t = make_tap(lp[0], lp[1], H, W);
                        if (!t.valid) continue;
                        const REAL w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw, w4 = t.lh * t.lw;
                        const int hl = t.h_low, wl = t.w_low, hh_ = hl + 1, wh = wl + 1;
                        const REAL* v1 = (hl >= 0 && wl >= 0) ? v + ((long)hl * W + wl) * row : 0;
                        const REAL* v2 = (hl >= 0 && wh <= W - 1) ? v + ((long)hl * W + wh) * row : 0;
                        const REAL* v3 = (hh_ <= H - 1 && wl >= 0) ? v + ((long)hh_ * W + wl) * row : 0;
                        const REAL* v4 = (hh_ <= H - 1 && wh <= W - 1) ? v + ((long)hh_ * W + wh) * row : 0;
                        const REAL aw = *ap;
                        for (int c = 0; c < D; ++c) {
                            const REAL val = w1 * (v1 ? v1[c] : 0) + w2 * (v2 ? v2[c] : 0) + w3 * (v3 ? v3[c] : 0) +
                                             w4 * (v4 ? v4[c] : 0);
                            o[c] += val * aw; /* cuh:290 */
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* grad_value, grad_loc, grad_attn are fully overwritten (zero-initialised here, as the host wrapper
 * does with at::zeros_like, ms_deform_attn_cuda.cu:121-123).  grad_value accumulation is done serially
 * per batch element, so the result is deterministic (the reference's atomics are not). */
int FN(oracle_msda_backward)(const REAL* value, const int32_t* shapes, const int32_t* level_start,
                             const REAL* loc, const REAL* attn, const REAL* grad_out, REAL* grad_value,
                             REAL* grad_loc, REAL* grad_attn, int N, int S, int M, int D, int L, int Lq, int P) {
    const long row = (long)M * D;
    memset(grad_value, 0, sizeof(REAL) * (size_t)N * S * row);
    memset(grad_loc, 0, sizeof(REAL) * (size_t)N * Lq * M * L * P * 2);
    memset(grad_attn, 0, sizeof(REAL) * (size_t)N * Lq * M * L * P);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < N; ++b) {
        for (int q = 0; q < Lq; ++q) {
            for (int m = 0; m < M; ++m) {
                const REAL* g = grad_out + (((long)b * Lq + q) * M + m) * D;
                const long sidx = ((long)b * Lq + q) * M + m;
                const REAL* lp = loc + sidx * L * P * 2;
                const REAL* ap = attn + sidx * L * P;
                REAL* glp = grad_loc + sidx * L * P * 2;
                REAL* gap = grad_attn + sidx * L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = shapes[2 * l], W = shapes[2 * l + 1];
                    const long base = ((long)b * S + level_start[l]) * row + (long)m * D;
                    const REAL* v = value + base;
                    REAL* gv = grad_value + base;
                    for (int p = 0; p < P; ++p, lp += 2, ++ap, glp += 2, ++gap) {
                        const tap_t t = make_tap(lp[0], lp[1], H, W);
                        if (!t.valid) continue; /* grads stay 0, cuh:361-368 */
                        const REAL w1 = t.hh * t.hw, w2 = t.hh * t.lw, w3 = t.lh * t.hw, w4 = t.lh * t.lw;
                        const int hl = t.h_low, wl = t.w_low, hh_ = hl + 1, wh = wl + 1;
                        const int ok1 = (hl >= 0 && wl >= 0), ok2 = (hl >= 0 && wh <= W - 1);
                        const int ok3 = (hh_ <= H - 1 && wl >= 0), ok4 = (hh_ <= H - 1 && wh <= W - 1);
                        const long o1 = ((long)hl * W + wl) * row, o2 = ((long)hl * W + wh) * row;
                        const long o3 = ((long)hh_ * W + wl) * row, o4 = ((long)hh_ * W + wh) * row;
                        const REAL aw = *ap;
                        REAL s_attn = 0, s_w = 0, s_h = 0; /* sums over channels, cuh:376-394 */
                        for (int c = 0; c < D; ++c) {
                            const REAL top = g[c];
                            const REAL tgv = top * aw; /* cuh:110 */
                            REAL ghw = 0, gww = 0;
                            REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                            if (ok1) { v1 = v[o1 + c]; ghw -= t.hw * v1; gww -= t.hh * v1; gv[o1 + c] += w1 * tgv; }
                            if (ok2) { v2 = v[o2 + c]; ghw -= t.lw * v2; gww += t.hh * v2; gv[o2 + c] += w2 * tgv; }
                            if (ok3) { v3 = v[o3 + c]; ghw += t.hw * v3; gww -= t.lh * v3; gv[o3 + c] += w3 * tgv; }
                            if (ok4) { v4 = v[o4 + c]; ghw += t.lw * v4; gww += t.lh * v4; gv[o4 + c] += w4 * tgv; }
                            const REAL val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                            s_attn += top * val;      /* cuh:156 */
                            s_w += W * gww * tgv;     /* cuh:157 */
                            s_h += H * ghw * tgv;     /* cuh:158 */
                        }
                        *gap = s_attn;
                        glp[0] = s_w;
                        glp[1] = s_h;
                    }
                }
            }
        }
    }
    return 0;
}
