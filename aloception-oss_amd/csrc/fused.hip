// Streaming epilogues of the layers either side of the attention op, each one pass over HBM:
//   alo_add_layernorm   out = LayerNorm(x + residual) * gamma + beta   [, out_pos = out + pos]
//   alo_bias_act        y   = act(x + bias[c] [+ residual])            channels-last rows, in place allowed
// Stock PyTorch runs these as 2-3 separate elementwise / normalisation kernels (and its LayerNorm kernel reaches a fifth of
// the HBM rate on (177784, 256) bf16 rows); here one wave owns one row of the normalisation, the row lives in registers,
// both reductions are wave-level DPP/shuffle reductions, and every byte is read and written once.
#include "common.hpp"

namespace alo {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 4;  // C <= 64 lanes * 4 elements * 4 chunks = 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const u32x2 x = *reinterpret_cast<const u32x2*>(p);
    v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xffff0000u);
    v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    u32x2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = o;
}

// One wave per row; lane i holds elements [256*k + 4*i, +4) of the row for k < CHUNKS (coalesced 8-/16-byte accesses).
template <typename T, int CHUNKS, bool HAS_RES, bool HAS_POS>
__global__ void __launch_bounds__(kThreads)
add_layernorm_kernel(const T* x, const T* res, const T* __restrict__ gamma, const T* __restrict__ beta, T* out,
                     const T* pos, T* out_pos, long rows, int C, float eps) {  // out may alias x / res: no restrict there
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (kThreads / 64);
    float g[CHUNKS][4], b[CHUNKS][4];
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const int c = 256 * k + 4 * lane;
        if (c < C) { load4(gamma + c, g[k]); load4(beta + c, b[k]); }
    }
    const float inv_c = 1.0f / (float)C;
    for (long r = wave; r < rows; r += nwaves) {
        const long base = r * C;
        float v[CHUNKS][4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k) {
            const int c = 256 * k + 4 * lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[k][i] = 0.f;
            if (c < C) {
                load4(x + base + c, v[k]);
                if constexpr (HAS_RES) {
                    float t[4];
                    load4(res + base + c, t);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[k][i] += t[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s += v[k][i];
            }
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[k][i] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < C) {
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (v[k][i] - mean) * rstd * g[k][i] + b[k][i];
                store4(out + base + c, y);
                if constexpr (HAS_POS) {
                    float p[4];
                    load4(pos + base + c, p);
                    // the sum is taken on the value the caller will see in `out` (rounded to T), as `out + pos` would be
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] = bf16_to_f32(f32_to_bf16(y[i]));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] += p[i];
                    store4(out_pos + base + c, y);
                }
            }
        }
    }
}

// y[r, c] = act(x[r, c] + bias[c] (+ residual[r, c])), 4 elements per thread, rows*C % 4 == 0 and C % 4 == 0.
template <typename T, bool HAS_RES, bool RELU>
__global__ void __launch_bounds__(kThreads)
bias_act_kernel(const T* x, const T* __restrict__ bias, const T* res, T* y,  // y may alias x
                long n4, int C) {
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const long e = i * 4;
        const int c = (int)(e % C);
        float v[4], bb[4];
        load4(x + e, v);
        load4(bias + c, bb);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += bb[k];
        if constexpr (HAS_RES) {
            float t[4];
            load4(res + e, t);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += t[k];
        }
        if constexpr (RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        store4(y + e, v);
    }
}

// out[n, m, s, :] = mask[n, s] ? 0 : value[n, s, m, :]   (pixel-major -> head-major, padding zeroed on the way).
// A workgroup moves 64 pixels: reads their (M*D)-element rows with 16-byte loads (fully coalesced), writes per head 64
// consecutive D-element rows (64*D*2 bytes contiguous).  16 B = 8 bf16 per thread.
__global__ void __launch_bounds__(kThreads)
value_head_major_kernel(const bf16_t* __restrict__ value, const unsigned char* __restrict__ mask,
                        bf16_t* __restrict__ out, int S, int M, int D) {
    const int n = blockIdx.y;
    const int s0 = blockIdx.x * 64;
    const int vec_per_row = M * D / 8;            // 16-byte vectors per pixel row
    const int vec_per_head = D / 8;
    const int total = 64 * vec_per_row;
    for (int i = threadIdx.x; i < total; i += kThreads) {
        // thread order follows the OUTPUT: (head, pixel, vector-in-head) so that stores are contiguous per head
        const int m = i / (64 * vec_per_head);
        const int rem = i - m * 64 * vec_per_head;
        const int sp = rem / vec_per_head, v = rem - sp * vec_per_head;
        const int s = s0 + sp;
        if (s >= S) continue;
        u32x4 x = *reinterpret_cast<const u32x4*>(value + ((size_t)n * S + s) * M * D + m * D + v * 8);
        if (mask != nullptr && mask[(size_t)n * S + s]) x = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(out + (((size_t)n * M + m) * S + s) * D + v * 8) = x;
    }
}

// ---- RAFT update block (fp32, NCHW planes of HW elements) --------------------------------------------------------------
// y[b, c, :] = act(x[b, c, :] + bias[c]); HW % 4 == 0; 4 elements per thread.
template <bool RELU>
__global__ void __launch_bounds__(kThreads)
bias_act_nchw_kernel(const float* x, const float* __restrict__ bias, float* y, long n4, int C, int HW4) {
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i / HW4) % C);
        float v[4];
        load4(x + i * 4, v);
        const float b = bias[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += b; if (RELU) v[k] = fmaxf(v[k], 0.f); }
        store4(y + i * 4, v);
    }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// GRU gates (alonet/raft/update.py:27-33): zr holds the pre-activations of the update gate z (channels [0,C)) and the
// reset gate r (channels [C,2C)) of ONE convolution over [h | x].  Writes z = sigmoid(.) back over its own slot and
// r * h into the first C channels of the [r*h | x] buffer.  h and rh are channel slices of (B, C + Cx, H, W) buffers:
// `hb` / `rb` are their batch strides in elements.
__global__ void __launch_bounds__(kThreads)
gru_gate_kernel(float* zr, const float* __restrict__ bias_zr, const float* h, float* rh, long n4, int C, int HW4,
                long hb, long rb) {
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const long plane = i / HW4;                 // b * C + c
        const int p4 = (int)(i - plane * HW4);
        const int b = (int)(plane / C), c = (int)(plane - (long)b * C);
        const long zoff = (((long)b * 2 * C + c) * HW4 + p4) * 4;
        const long roff = zoff + (long)C * HW4 * 4;
        const long hoff = (long)b * hb + ((long)c * HW4 + p4) * 4;
        const long rhoff = (long)b * rb + ((long)c * HW4 + p4) * 4;
        float z[4], r[4], hv[4];
        load4(zr + zoff, z);
        load4(zr + roff, r);
        load4(h + hoff, hv);
        const float bz = bias_zr[c], br = bias_zr[C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z[k] = sigmoidf_(z[k] + bz);
            r[k] = sigmoidf_(r[k] + br) * hv[k];
        }
        store4(zr + zoff, z);
        store4(rh + rhoff, r);
    }
}

// h <- (1 - z) * h + z * tanh(q + bias_q), in place in the [h | x] buffer; optionally also to a contiguous copy `net`.
__global__ void __launch_bounds__(kThreads)
gru_update_kernel(const float* q, const float* __restrict__ bias_q, const float* zr, float* h, float* net, long n4, int C,
                  int HW4, long hb) {
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const long plane = i / HW4;
        const int p4 = (int)(i - plane * HW4);
        const int b = (int)(plane / C), c = (int)(plane - (long)b * C);
        const long zoff = (((long)b * 2 * C + c) * HW4 + p4) * 4;
        const long hoff = (long)b * hb + ((long)c * HW4 + p4) * 4;
        float qv[4], z[4], hv[4];
        load4(q + i * 4, qv);
        load4(zr + zoff, z);
        load4(h + hoff, hv);
        const float bq = bias_q[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = (1.0f - z[k]) * hv[k] + z[k] * tanhf(qv[k] + bq);
        store4(h + hoff, hv);
        if (net != nullptr) store4(net + i * 4, hv);
    }
}

// ---- sine positional encoding of the flattened pyramid (alonet/transformers/position_encoding.py:29-72) ---------------------
// Pass 1: per (level, image) the cumulative count of un-padded pixels along y and along x, centred / normalised the way the
// module does it, as (ey, ex) per pixel.  Work items are WAVES: one per 64 columns (a lane walks its column: coalesced byte loads,
// two passes — total, then running count) and one per image row (a lane owns a run of ceil(W / 64) pixels; wave-level inclusive
// scan of the run totals).  grid = (workers, L, B); the 4 waves of worker w take items 4 w + wave, + 4 * workers, ...
__global__ void __launch_bounds__(kThreads)
pos_prefix_kernel(const unsigned char* __restrict__ mask, const int32_t* __restrict__ shapes,
                  const int32_t* __restrict__ lstart, float* __restrict__ emb, int S, int normalize, int center,
                  float scale, float eps) {
    const int l = blockIdx.y, b = blockIdx.z;
    const int H = shapes[2 * l], W = shapes[2 * l + 1], start = lstart[l];
    const unsigned char* mk = mask + (size_t)b * S + start;
    float* e = emb + ((size_t)b * S + start) * 2;
    const float shift = (normalize && center) ? 0.5f : 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col_items = (W + 63) / 64, items = col_items + H;
    for (int item = blockIdx.x * (kThreads / 64) + wave; item < items; item += gridDim.x * (kThreads / 64)) {
        if (item < col_items) {   // ---- y_embed of 64 columns ----
            const int x = item * 64 + lane;
            if (x < W) {
                float cum = 0.f;
#pragma unroll 4
                for (int y = 0; y < H; ++y) cum += mk[y * W + x] ? 0.f : 1.f;
                const float denom = (cum - shift) + eps;  // y_embed[:, -1:, :] + eps
                float run = 0.f;
#pragma unroll 4
                for (int y = 0; y < H; ++y) {
                    run += mk[y * W + x] ? 0.f : 1.f;
                    e[(y * W + x) * 2] = normalize ? (run - shift) / denom * scale : run;
                }
            }
        } else {                  // ---- x_embed of one row ----
            const int y = item - col_items;
            const int per = (W + 63) / 64, x0 = lane * per, x1 = min(W, x0 + per);
            float mine = 0.f;
            for (int x = x0; x < x1; ++x) mine += mk[y * W + x] ? 0.f : 1.f;
            float incl = mine;   // inclusive scan of the lanes' run totals (exact: small integers)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const float total = __shfl(incl, 63, 64);
            const float denom = (total - shift) + eps;
            float run = incl - mine;
            for (int x = x0; x < x1; ++x) {
                run += mk[y * W + x] ? 0.f : 1.f;
                e[(y * W + x) * 2 + 1] = normalize ? (run - shift) / denom * scale : run;
            }
        }
    }
}

// Pass 2: out[b, s, c] = (c < F ? f(ey) : f(ex)) + level_embed[level(s), c],  f = sin on even frequencies, cos on odd ones,
// argument = embed / dim_t[c mod F].  4 channels per thread.
template <typename T>
__global__ void __launch_bounds__(kThreads)
pos_generate_kernel(const float* __restrict__ emb, const float* __restrict__ dim_t, const T* __restrict__ level_embed,
                    const int32_t* __restrict__ lstart, T* __restrict__ out, long n4, int S, int L, int F) {
    const int C = 2 * F, c4 = C / 4;
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const long pix = i / c4;                 // b * S + s
        const int c0 = (int)(i - pix * c4) * 4;
        const int s = (int)(pix % S);
        int l = 0;
        while (l + 1 < L && s >= lstart[l + 1]) ++l;
        const float e = emb[pix * 2 + (c0 < F ? 0 : 1)];
        float le[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
        if (level_embed != nullptr) load4(level_embed + (size_t)l * C + c0, le);
        // channels (2 i, 2 i + 1) are (sin, cos) of the SAME argument (dim_t[2 i] == dim_t[2 i + 1]): one sincosf per pair
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            const int f = (c0 + k) % F;   // even: c0 % 4 == 0 and F % 4 == 0
            const float p = e / dim_t[f];
            float sn, cs;
            sincosf(p, &sn, &cs);
            v[k] = sn + le[k];
            v[k + 1] = (dim_t[f + 1] == dim_t[f] ? cs : cosf(e / dim_t[f + 1])) + le[k + 1];
        }
        store4(out + pix * C + c0, v);
    }
}

template <typename K>
int launch(K kernel, unsigned blocks, hipStream_t stream, const char* what, void** args) {
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), dim3(blocks), dim3(kThreads), args, 0, stream);
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return check_launch(what);
}

template <typename T>
int add_layernorm_t(const void* x, const void* res, const void* gamma, const void* beta, void* out, const void* pos,
                    void* out_pos, long rows, int C, float eps, hipStream_t stream) {
    const int chunks = (C + 255) / 256;
    long blocks = (rows + 3) / 4;
    if (blocks > 256L * 32) blocks = 256L * 32;
    void* args[] = {&x, &res, &gamma, &beta, &out, &pos, &out_pos, &rows, &C, &eps};
    const char* what = "alo_add_layernorm";
#define ALO_LN_CASE(CH)                                                                                            \
    if (chunks == CH) {                                                                                            \
        if (res && pos) return launch(add_layernorm_kernel<T, CH, true, true>, (unsigned)blocks, stream, what, args);   \
        if (res) return launch(add_layernorm_kernel<T, CH, true, false>, (unsigned)blocks, stream, what, args);         \
        if (pos) return launch(add_layernorm_kernel<T, CH, false, true>, (unsigned)blocks, stream, what, args);         \
        return launch(add_layernorm_kernel<T, CH, false, false>, (unsigned)blocks, stream, what, args);                 \
    }
    ALO_LN_CASE(1) ALO_LN_CASE(2) ALO_LN_CASE(3) ALO_LN_CASE(4)
#undef ALO_LN_CASE
    return fail(ALO_ERR_UNSUPPORTED, "alo_add_layernorm: C = %d is above %d", C, 256 * kMaxChunks);
}

template <typename T>
int bias_act_t(const void* x, const void* bias, const void* res, void* y, long rows, int C, int relu, hipStream_t stream) {
    long n4 = rows * C / 4;
    long blocks = (n4 + kThreads - 1) / kThreads;
    if (blocks > 256L * 64) blocks = 256L * 64;
    void* args[] = {&x, &bias, &res, &y, &n4, &C};
    const char* what = "alo_bias_act";
    if (res) {
        if (relu) return launch(bias_act_kernel<T, true, true>, (unsigned)blocks, stream, what, args);
        return launch(bias_act_kernel<T, true, false>, (unsigned)blocks, stream, what, args);
    }
    if (relu) return launch(bias_act_kernel<T, false, true>, (unsigned)blocks, stream, what, args);
    return launch(bias_act_kernel<T, false, false>, (unsigned)blocks, stream, what, args);
}

}  // namespace
}  // namespace alo

using namespace alo;

extern "C" int alo_add_layernorm(const void* x, const void* residual, const void* gamma, const void* beta, void* out,
                                 const void* pos, void* out_pos, long rows, int C, float eps, int dtype, void* stream) {
    ALO_REQUIRE(x && gamma && beta && out, ALO_ERR_INVALID_ARGUMENT, "alo_add_layernorm: null pointer argument");
    ALO_REQUIRE((pos == nullptr) == (out_pos == nullptr), ALO_ERR_INVALID_ARGUMENT,
                "alo_add_layernorm: pos and out_pos go together");
    ALO_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_add_layernorm: rows must be positive and C a positive multiple of 4 (rows=%ld C=%d)", rows, C);
    const uintptr_t all = (uintptr_t)x | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out |
                          (uintptr_t)pos | (uintptr_t)out_pos;
    ALO_REQUIRE((all & 15) == 0, ALO_ERR_INVALID_ARGUMENT, "alo_add_layernorm: pointers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ALO_F32) return add_layernorm_t<float>(x, residual, gamma, beta, out, pos, out_pos, rows, C, eps, s);
    if (dtype == ALO_BF16) return add_layernorm_t<bf16_t>(x, residual, gamma, beta, out, pos, out_pos, rows, C, eps, s);
    return fail(ALO_ERR_UNSUPPORTED, "alo_add_layernorm: dtype %d (F32 and BF16 are supported)", dtype);
}

extern "C" int alo_bias_act(const void* x, const void* bias, const void* residual, void* y, long rows, int C, int relu,
                            int dtype, void* stream) {
    ALO_REQUIRE(x && bias && y, ALO_ERR_INVALID_ARGUMENT, "alo_bias_act: null pointer argument");
    ALO_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_bias_act: rows must be positive and C a positive multiple of 4 (rows=%ld C=%d)", rows, C);
    const uintptr_t all = (uintptr_t)x | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)y;
    ALO_REQUIRE((all & 15) == 0, ALO_ERR_INVALID_ARGUMENT, "alo_bias_act: pointers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ALO_F32) return bias_act_t<float>(x, bias, residual, y, rows, C, relu, s);
    if (dtype == ALO_BF16) return bias_act_t<bf16_t>(x, bias, residual, y, rows, C, relu, s);
    return fail(ALO_ERR_UNSUPPORTED, "alo_bias_act: dtype %d (F32 and BF16 are supported)", dtype);
}

extern "C" int alo_value_head_major(const void* value, const void* padding_mask, void* out, int N, int S, int M, int D,
                                    int dtype, void* stream) {
    ALO_REQUIRE(value && out, ALO_ERR_INVALID_ARGUMENT, "alo_value_head_major: null pointer argument");
    ALO_REQUIRE(N > 0 && S > 0 && M > 0 && D > 0 && D % 8 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_value_head_major: dimensions must be positive and D a multiple of 8 (N=%d S=%d M=%d D=%d)", N, S, M, D);
    ALO_REQUIRE(dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_value_head_major: bf16 only (dtype %d)", dtype);
    ALO_REQUIRE((((uintptr_t)value | (uintptr_t)out) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_value_head_major: pointers must be 16-byte aligned");
    void* args[] = {&value, &padding_mask, &out, &S, &M, &D};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(value_head_major_kernel), dim3((S + 63) / 64, N),
                                   dim3(kThreads), args, 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_value_head_major: %s", hipGetErrorString(e));
    return check_launch("alo_value_head_major");
}

static unsigned stream_blocks(long n4) {
    long blocks = (n4 + kThreads - 1) / kThreads;
    return (unsigned)(blocks > 256L * 64 ? 256L * 64 : blocks);
}

extern "C" int alo_bias_act_nchw(const float* x, const float* bias, float* y, int B, int C, int HW, int relu, void* stream) {
    ALO_REQUIRE(x && bias && y, ALO_ERR_INVALID_ARGUMENT, "alo_bias_act_nchw: null pointer argument");
    ALO_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_bias_act_nchw: dimensions must be positive and H*W a multiple of 4 (B=%d C=%d HW=%d)", B, C, HW);
    ALO_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_bias_act_nchw: pointers must be 16-byte aligned");
    long n4 = (long)B * C * HW / 4;
    int hw4 = HW / 4;
    void* args[] = {&x, &bias, &y, &n4, &C, &hw4};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (relu) return launch(bias_act_nchw_kernel<true>, stream_blocks(n4), s, "alo_bias_act_nchw", args);
    return launch(bias_act_nchw_kernel<false>, stream_blocks(n4), s, "alo_bias_act_nchw", args);
}

extern "C" int alo_gru_gate(float* zr, const float* bias_zr, const float* h, float* rh, int B, int C, int HW,
                            long h_batch_stride, long rh_batch_stride, void* stream) {
    ALO_REQUIRE(zr && bias_zr && h && rh, ALO_ERR_INVALID_ARGUMENT, "alo_gru_gate: null pointer argument");
    ALO_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0 && h_batch_stride % 4 == 0 && rh_batch_stride % 4 == 0,
                ALO_ERR_INVALID_ARGUMENT, "alo_gru_gate: H*W and the batch strides must be multiples of 4 (B=%d C=%d HW=%d)",
                B, C, HW);
    ALO_REQUIRE((((uintptr_t)zr | (uintptr_t)h | (uintptr_t)rh) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_gru_gate: pointers must be 16-byte aligned");
    long n4 = (long)B * C * HW / 4;
    int hw4 = HW / 4;
    void* args[] = {&zr, &bias_zr, &h, &rh, &n4, &C, &hw4, &h_batch_stride, &rh_batch_stride};
    return launch(gru_gate_kernel, stream_blocks(n4), static_cast<hipStream_t>(stream), "alo_gru_gate", args);
}

extern "C" int alo_gru_update(const float* q, const float* bias_q, const float* zr, float* h, float* net, int B, int C,
                              int HW, long h_batch_stride, void* stream) {
    ALO_REQUIRE(q && bias_q && zr && h, ALO_ERR_INVALID_ARGUMENT, "alo_gru_update: null pointer argument");
    ALO_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0 && h_batch_stride % 4 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_gru_update: H*W and the batch stride must be multiples of 4 (B=%d C=%d HW=%d)", B, C, HW);
    ALO_REQUIRE((((uintptr_t)q | (uintptr_t)zr | (uintptr_t)h | (uintptr_t)net) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_gru_update: pointers must be 16-byte aligned");
    long n4 = (long)B * C * HW / 4;
    int hw4 = HW / 4;
    void* args[] = {&q, &bias_q, &zr, &h, &net, &n4, &C, &hw4, &h_batch_stride};
    return launch(gru_update_kernel, stream_blocks(n4), static_cast<hipStream_t>(stream), "alo_gru_update", args);
}

extern "C" int alo_pos_sine_flat(const void* padding_mask, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                 const float* dim_t, const void* level_embed, void* out, float* workspace, int B, int S,
                                 int L, int num_pos_feats, int normalize, int center, float scale, float eps, int dtype,
                                 void* stream) {
    ALO_REQUIRE(padding_mask && spatial_shapes && level_start_index && dim_t && out && workspace, ALO_ERR_INVALID_ARGUMENT,
                "alo_pos_sine_flat: null pointer argument");
    ALO_REQUIRE(B > 0 && S > 0 && L > 0 && num_pos_feats > 0 && num_pos_feats % 4 == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_pos_sine_flat: sizes must be positive and num_pos_feats a multiple of 4 (B=%d S=%d L=%d F=%d)", B, S, L,
                num_pos_feats);
    ALO_REQUIRE(dtype == ALO_F32 || dtype == ALO_BF16, ALO_ERR_UNSUPPORTED, "alo_pos_sine_flat: dtype %d", dtype);
    ALO_REQUIRE((((uintptr_t)out | (uintptr_t)level_embed) & 15) == 0, ALO_ERR_INVALID_ARGUMENT,
                "alo_pos_sine_flat: out / level_embed must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        void* args[] = {&padding_mask, &spatial_shapes, &level_start_index, &workspace, &S, &normalize, &center, &scale, &eps};
        hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(pos_prefix_kernel), dim3(32, L, B), dim3(kThreads), args, 0, st);
        if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "alo_pos_sine_flat: %s", hipGetErrorString(e));
    }
    long n4 = (long)B * S * (2 * num_pos_feats) / 4;
    void* args[] = {&workspace, &dim_t, &level_embed, &level_start_index, &out, &n4, &S, &L, &num_pos_feats};
    if (dtype == ALO_F32) return launch(pos_generate_kernel<float>, stream_blocks(n4), st, "alo_pos_sine_flat", args);
    return launch(pos_generate_kernel<bf16_t>, stream_blocks(n4), st, "alo_pos_sine_flat", args);
}
