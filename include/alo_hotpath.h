/*
 * alo_hotpath.h — C ABI of the MI355X (gfx950) dense-vision hot path of aloception.
 *
 * One shared library (libalo_hotpath.so, built from aloception-oss_amd/csrc/ with hipcc) exports exactly the entry
 * points the reference's own binding for this path would bind.  Plain pointers and sizes only: no torch / ATen types.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   alo_msda_forward   <- alonet_custom::ms_deform_attn_forward   alonet/deformable_detr/ops/src/vision.cpp:21-24,
 *                          ms_deform_attn.h:20-39, cuda/ms_deform_attn_cuda.cu:20-80, cuda/ms_deform_im2col_cuda.cuh:237-299
 *   alo_msda_forward_fused <- the elementwise prologue of MSDeformAttn.forward + the op   ops/modules/ms_deform_attn.py:119-153
 *   alo_msda_backward  <- alonet_custom::ms_deform_attn_backward  ms_deform_attn.h:41-62, cuda/ms_deform_attn_cuda.cu:83-153,
 *                          cuda/ms_deform_im2col_cuda.cuh:87-234,301-920
 *   alo_corr_build     <- CorrBlock.__init__ / CorrBlock.corr     alonet/raft/corr.py:13-27,52-60
 *   alo_corr_lookup    <- CorrBlock.__call__ + bilinear_sampler   alonet/raft/corr.py:29-50, alonet/raft/utils/utils.py:5-19
 *
 * Conventions (all entry points):
 *   - every data pointer is a DEVICE pointer on the current HIP device, densely packed ("contiguous") in the layout
 *     documented per function; the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and the call returns
 *     immediately — same contract as the reference op, which enqueues on the current torch stream;
 *   - outputs are fully overwritten (the caller does not need to zero them);
 *   - return value: ALO_OK (0) or an alo_status_t error; alo_last_error() gives a thread-local message.  Argument
 *     errors are detected before anything is enqueued;
 *   - stateless and re-entrant; safe to call from several host threads on different streams.
 */
#ifndef ALO_HOTPATH_H
#define ALO_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALO_HOTPATH_ABI_VERSION 2   /* 2: alo_corr_lookup_conv1x1[_kpad] removed (round 5), alo_msda_backward_path added */

typedef enum alo_status {
    ALO_OK = 0,
    ALO_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive dimension, misaligned pointer ...           */
    ALO_ERR_UNSUPPORTED = 2,      /* dtype combination or size outside what the kernels are built for       */
    ALO_ERR_LAUNCH = 3            /* the HIP runtime refused the launch (message carries hipGetErrorString)  */
} alo_status_t;

/* Element types.  For MSDA the reference dispatches float and double (AT_DISPATCH_FLOATING_TYPES,
 * ms_deform_attn_cuda.cu:64,134); bf16 storage with fp32 arithmetic is this library's addition. */
typedef enum alo_dtype {
    ALO_F32 = 0,
    ALO_F64 = 1,
    ALO_BF16 = 2
} alo_dtype_t;

/* ABI version of the loaded library (== ALO_HOTPATH_ABI_VERSION it was built with). */
int alo_abi_version(void);

/* Message of the last error raised on the calling thread ("" if none).  Never NULL. */
const char* alo_last_error(void);

/*
 * Multi-scale deformable attention, forward.
 *
 *   out[b,q,m,c] = sum_{l<L} sum_{p<P} attn[b,q,m,l,p] * bilinear(value_l[b,:,m,c]; x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5)
 *
 * A sample is skipped unless y > -1 && x > -1 && y < H_l && x < W_l; each of its four corners is bounds-checked on its
 * own (zero padding, i.e. grid_sample(align_corners=False, padding_mode="zeros") semantics).
 *
 *   value               (N, S, M, D)        value_dtype          S = sum_l H_l*W_l
 *   spatial_shapes      (L, 2) int32        [H_l, W_l]           (device memory, as in the reference)
 *   level_start_index   (L,)   int32        first row of level l inside S
 *   sampling_loc        (N, Lq, M, L, P, 2) loc_dtype            last dim = (x, y), normalised to [0,1] of the padded map
 *   attn_weight         (N, Lq, M, L, P)    loc_dtype
 *   out                 (N, Lq, M*D)        value_dtype          m-major, c-minor
 *
 * dtype pairs (value_dtype, loc_dtype): (F32,F32) (F64,F64) (BF16,F32).  Arithmetic is fp64 for F64 and fp32
 * otherwise (bf16 locations would cost a third of a pixel on a 167-wide map, so they are not offered).  Any D >= 1 is accepted; D % (16 / sizeof(value element)) == 0 with 16-byte aligned `value`/`out`
 * takes the vectorised path.  L <= 32.  N*S*M*D*sizeof(element) per batch item must stay below 3 GiB.
 * The reference's `im2col_step` is a scheduling hint with no effect on results and has no counterpart here.
 */
int alo_msda_forward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                     const void* sampling_loc, const void* attn_weight, void* out,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     int value_dtype, int loc_dtype, void* stream);

/*
 * Multi-scale deformable attention, forward, with MSDeformAttn's prologue fused in (an extension: the reference has no
 * such entry point — it evaluates these steps as separate PyTorch ops, alonet/deformable_detr/ops/modules/ms_deform_attn.py:119-133):
 *
 *   attn[b,q,m,:]  = softmax over the L*P entries of attn_logits[b,q,m,:]
 *   loc[b,q,m,l,p] = ref[b,q,l,0:2] + sampling_offsets[b,q,m,l,p,:] / (W_l, H_l)                       (ref_dim == 2)
 *                  = ref[b,q,l,0:2] + sampling_offsets[b,q,m,l,p,:] / P * ref[b,q,l,2:4] * 0.5          (ref_dim == 4)
 *   out            = alo_msda_forward(value, ..., loc, attn)
 *
 *   sampling_offsets    (N, Lq, M, L, P, 2)   value_dtype      raw output of the sampling_offsets linear layer
 *   attn_logits         (N, Lq, M, L*P)       value_dtype      raw output of the attention_weights linear layer
 *   reference_points    (N, Lq, L, ref_dim)   F64 when value_dtype is F64, F32 otherwise
 *
 * Softmax and location arithmetic run in fp32 (fp64 for F64) inside the kernel's descriptor stage; nothing but `out`
 * is written.  Inference path only (the gradient path uses alo_msda_forward / alo_msda_backward).
 */
int alo_msda_forward_fused(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                           const void* sampling_offsets, const void* attn_logits, const void* reference_points,
                           void* out, int N, int S, int M, int D, int L, int Lq, int P, int ref_dim,
                           int value_dtype, void* stream);

/*
 * Head-major variant of alo_msda_forward_fused (extension).  `value_hm` is (N, M, S, D): what alo_value_head_major
 * writes from the boundary layout.  Everything else as above; results are bit-identical to alo_msda_forward_fused on
 * the same data.  bf16, L = P = 4, D % 8 == 0, D <= 32 only (ALO_ERR_UNSUPPORTED otherwise).  Why it exists: a head's
 * row is 64 bytes = half an L1 line; head-major rows of neighbouring pixels share lines, pixel-major ones never do.
 */
int alo_msda_forward_fused_hm(const void* value_hm, const int32_t* spatial_shapes, const int32_t* level_start_index,
                              const void* sampling_offsets, const void* attn_logits, const void* reference_points,
                              void* out, int N, int S, int M, int D, int L, int Lq, int P, int ref_dim,
                              int value_dtype, void* stream);
/*
 * Same, with the RAW offsets and logits of a query allowed to sit in rows of a wider buffer: `sampling_offsets` points at the query-0
 * offsets, consecutive queries are offsets_row_elems elements apart (>= M*L*P*2), likewise `attn_logits` / logits_row_elems (>= M*L*P).
 * Lets `sampling_offsets` and `attention_weights` (ms_deform_attn.py:119-121) be ONE GEMM with 3*M*L*P output columns.
 */
int alo_msda_forward_fused_hm_rows(const void* value_hm, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                   const void* sampling_offsets, const void* attn_logits, long offsets_row_elems,
                                   long logits_row_elems, const void* reference_points, void* out, int N, int S, int M, int D,
                                   int L, int Lq, int P, int ref_dim, int value_dtype, void* stream);

/*
 * Same as alo_msda_forward_fused_hm_rows, with the COARSE pyramid levels of every (image, head) slab kept resident in LDS (extension;
 * csrc/msda.hip: msda_fwd_bf16_resident_kernel).  `host_spatial_shapes` is a HOST copy of spatial_shapes (L x 2 int32, [H, W] per
 * level): it decides whether levels 2-3 may be resident (their rows must fit in a CU's LDS next to the waves' work areas: about
 * 1 400 pixels, i.e. frames up to ~1333 x 800 at strides 8-64), fixes the layout of the LDS image and sizes the grid; the kernel
 * compares it with the device copy and serves every level through the ordinary buffer path if the two disagree, so a stale host copy
 * costs time, never correctness.  D must be 32.
 * `policy`: ALO_RESIDENT_AUTO takes the resident kernel only where it is the faster one — launches with at least one 16-query run
 * per wave of the chip (N * M * ceil(Lq / 16) >= CUs * 12; measured 0.77-0.86 of the plain kernel's time above that, 1.2-1.6 x
 * below) — and the plain head-major kernel otherwise (small frames, the decoder's 300 queries): callers may use it unconditionally.
 * ALO_RESIDENT_ALWAYS takes the resident kernel wherever it can run (tests; callers that know better).  Results are bit-identical
 * to alo_msda_forward_fused_hm either way (same products, same order of the sum).
 * alo_msda_resident_levels reports what a launch of these dimensions would do under `policy`: 2 (levels 2-3 resident) or 0 (plain).
 */
#define ALO_RESIDENT_AUTO 0
#define ALO_RESIDENT_ALWAYS 1
int alo_msda_forward_fused_hm_resident(const void* value_hm, const int32_t* spatial_shapes, const int32_t* level_start_index,
                                       const void* sampling_offsets, const void* attn_logits, long offsets_row_elems,
                                       long logits_row_elems, const void* reference_points, void* out, int N, int S, int M, int D,
                                       int L, int Lq, int P, int ref_dim, int value_dtype, const int32_t* host_spatial_shapes,
                                       int policy, void* stream);
int alo_msda_resident_levels(const int32_t* host_spatial_shapes, int N, int S, int M, int L, int Lq, int policy);

/*
 * value (N, S, M, D) -> out (N, M, S, D), rows of padded pixels zeroed (padding_mask (N, S) uint8/bool, nullable):
 * MSDeformAttn's `value.masked_fill(input_padding_mask[..., None], 0)` (ms_deform_attn.py:112-113) and the re-layout
 * in one pass.  bf16, D % 8 == 0.
 */
int alo_value_head_major(const void* value, const void* padding_mask, void* out, int N, int S, int M, int D,
                         int dtype, void* stream);

/*
 * Multi-scale deformable attention, backward (gradients of the forward above w.r.t. value, sampling_loc, attn_weight).
 *
 *   grad_out            (N, Lq, M*D)        value_dtype
 *   grad_value          (N, S, M, D)        grad dtype           zeroed by this call, then accumulated with atomics
 *   grad_sampling_loc   (N, Lq, M, L, P, 2) grad dtype           fully written (0 for skipped samples)
 *   grad_attn_weight    (N, Lq, M, L, P)    grad dtype
 *
 * grad dtype is F64 when value_dtype is F64 and F32 otherwise (bf16 storage accumulates its gradients in fp32; the
 * caller narrows afterwards).  Supported (value_dtype, loc_dtype): (F32,F32) (F64,F64) (BF16,F32).
 * grad_value is accumulated with hardware floating-point atomics, so — exactly like the reference — its low-order
 * bits depend on scheduling.
 */
int alo_msda_backward(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                      const void* sampling_loc, const void* attn_weight, const void* grad_out,
                      void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      int value_dtype, int loc_dtype, void* stream);

/*
 * The same operation with a scheduling hint: `host_spatial_shapes` is a HOST copy of spatial_shapes (L x 2 int32, may be NULL).
 * When the queries are the pyramid's own pixels (Lq == S, the encoder's self-attention) the D = 32, L = P = 4 launches (fp32 or
 * bf16 values) then group them as 16x16 blocks of their level, sort a block's sampling corners by pixel on chip and issue ONE atomic
 * row per touched pixel per block (msda_bwd_wide.hip); without the hint fp32 takes 4x4 tiles of 16 consecutive queries.  Results do not depend on
 * the hint (up to the order of the floating-point additions, which atomics leave undefined anyway); alo_msda_backward is this
 * call with a NULL hint.
 */
int alo_msda_backward_hinted(const void* value, const int32_t* spatial_shapes, const int32_t* level_start_index,
                             const void* sampling_loc, const void* attn_weight, const void* grad_out,
                             void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                             int N, int S, int M, int D, int L, int Lq, int P,
                             int value_dtype, int loc_dtype, const int32_t* host_spatial_shapes, void* stream);

/*
 * Which kernel alo_msda_backward_hinted takes for a launch of these dimensions (pointers assumed 16-byte aligned), without
 * enqueuing anything: ALO_MSDA_BWD_WIDE (16x16 query blocks sorted on chip, one atomic row per touched pixel per block: fp32 / bf16
 * values, D = 32, L = P = 4, Lq == S and a host copy of the shapes), ALO_MSDA_BWD_TILED (4x4 query tiles on the fp32 matrix cores:
 * fp32, D = 32, L = P = 4), ALO_MSDA_BWD_GENERIC (one atomic row per corner: everything else); -1 for an unsupported dtype pair.
 */
#define ALO_MSDA_BWD_GENERIC 0
#define ALO_MSDA_BWD_TILED 1
#define ALO_MSDA_BWD_WIDE 2
int alo_msda_backward_path(int N, int S, int M, int D, int L, int Lq, int P, int value_dtype, int loc_dtype,
                           const int32_t* host_spatial_shapes);

/* Size of level l of the correlation pyramid of an (H, W) feature grid: level 0 = (H, W), level l+1 = floor(level l / 2)
 * (F.avg_pool2d(2, stride=2), corr.py:25-27). */
void alo_corr_level_shape(int H, int W, int level, int* h_out, int* w_out);

/* Bytes of scratch alo_corr_build needs (fp16-split copies of the feature maps, one power of two per pixel of each, fmap2's
 * per-item magnitude; the 2x2-average chain of fmap2 for pyramids deeper than 3 levels). */
size_t alo_corr_build_workspace_bytes(int B, int C, int H, int W, int num_levels);

/*
 * RAFT all-pairs correlation pyramid.
 *
 *   level_0[b*HW + i, 0, y, x] = <fmap1[b,:,i], fmap2[b,:,y*W + x]> / sqrt(C)
 *   level_{l+1}                = avg_pool2d(level_l, 2, stride 2)        over the last two dims
 *
 * Level 0 is one dense contraction on the fp16 matrix cores at fp32 accuracy (every PIXEL's feature vector scaled by a power of
 * two taken from its own largest magnitude, every feature split into two fp16 terms hi + lo = 22 significant bits, the cross
 * products hi*hi, hi*lo, lo*hi accumulated in fp32 and the two powers of two undone exactly per entry: error < 2^-21 relative
 * per product at any input magnitude, i.e. an entry is accurate relative to sum_c |f1_ci f2_cj| like an fp32 dot product, however
 * dim its pixels are next to the brightest of the item; non-finite features propagate to the rows / columns the reference makes
 * non-finite);
 * levels 1 and 2 are pooled from the accumulators in the same launch; deeper levels are the same contraction against the
 * 2x2-average chain of fmap2 (average pooling commutes with the inner product).  Results agree with the reference's
 * matmul + avg_pool2d chain to fp32 rounding.
 *
 *   fmap1, fmap2   (B, C, H, W) float32
 *   levels         HOST array of num_levels DEVICE pointers; levels[l] is (B*H*W, 1, h_l, w_l) float32, fully written
 *   workspace      device scratch of alo_corr_build_workspace_bytes(...) bytes, 16-byte aligned (the split operands)
 *   1 <= num_levels <= 8.  Levels whose h_l or w_l is 0 are rejected (ALO_ERR_INVALID_ARGUMENT).
 */
int alo_corr_build(const float* fmap1, const float* fmap2, float* const* levels, void* workspace,
                   size_t workspace_bytes, int B, int C, int H, int W, int num_levels, void* stream);

/*
 * RAFT windowed pyramid lookup.
 *
 *   out[b, l*(2r+1)^2 + a*(2r+1) + c, y, x] = bilinear_{align_corners=True, zeros}( level_l[b*HW + y*W + x] ;
 *                                                 px = coords[b,0,y,x] / 2^l + (a - r),  py = coords[b,1,y,x] / 2^l + (c - r) )
 *
 * i.e. the first window axis offsets x and the second offsets y (the reference's meshgrid(dy, dx) ordering).
 * Coordinates take the reference's round trip px -> 2*px/(w_l-1) - 1 -> ((g+1)/2)*(w_l-1) in fp32.
 *
 *   levels   HOST array of num_levels DEVICE pointers as produced by alo_corr_build
 *   coords   (B, 2, H, W) float32, channel 0 = x, 1 = y, in pixels of the (H, W) grid
 *   out      (B, num_levels*(2r+1)^2, H, W) float32, fully written
 *   0 <= radius <= 7,  1 <= num_levels <= 8.  Every level must have h_l, w_l >= 2 (the reference returns NaN otherwise).
 */
int alo_corr_lookup(const float* const* levels, const float* coords, float* out,
                    int B, int H, int W, int radius, int num_levels, void* stream);

/*
 * Backward of alo_corr_lookup with respect to the pyramid: the exact adjoint of the lookup, ACCUMULATED into gradient maps.
 * The reference's CorrBlock is plain torch code (alonet/raft/corr.py:29-50: bilinear_sampler = F.grid_sample) that autograd
 * differentiates; RAFT detaches the coordinates before every lookup (alonet/raft/raft.py:186), so this is the gradient a RAFT
 * training step needs from the block.  (The gradient with respect to the coordinates is a kernel of its own:
 * alo_corr_lookup_backward_coords below.)
 *
 *   grad_levels[l][b*HW + y*W + x, :, :] += sum over the window taps (a, c) of
 *                                           grad_out[b, l*(2r+1)^2 + a*(2r+1) + c, y, x] * bilinear weights of that tap
 *
 *   grad_levels  HOST array of num_levels DEVICE pointers, each (B*H*W, 1, h_l, w_l) float32 like the pyramid: read-modify-write
 *                (zero them before the first of the lookups whose gradients are to be summed; successive calls on one stream add up)
 *   coords       (B, 2, H, W) float32, the coordinates the forward lookup was given
 *   grad_out     (B, num_levels*(2r+1)^2, H, W) float32, contiguous
 * A query's window lies in its own (h_l, w_l) map on every level, so no two threads ever add to the same element: plain loads and
 * stores, deterministic.  Same limits as alo_corr_lookup.
 */
int alo_corr_lookup_backward(float* const* grad_levels, const float* coords, const float* grad_out,
                             int B, int H, int W, int radius, int num_levels, void* stream);

/*
 * Backward of alo_corr_lookup with respect to the COORDINATES (grid_sample's gradient with respect to the grid chained through the
 * reference's coordinate arithmetic, corr.py:29-50 / utils.py:5-20; RAFT itself detaches the coordinates, raft.py:186):
 *
 *   grad_coords_levels[b, l, 0, y, x] = 2^-l * sum over taps (a, c) of grad_out[b, l*(2r+1)^2 + a*(2r+1) + c, y, x] * d tap / d px
 *   grad_coords_levels[b, l, 1, y, x] = the same with d tap / d py
 *
 * with the bilinear interpolant's one-sided derivatives of the cell a tap falls in (values outside the map count as zeros, as in
 * grid_sample).  One (2, H, W) map per level, fully written; the gradient of `coords` is their sum over l (the caller adds them).
 *
 *   levels              HOST array of num_levels DEVICE pointers: the pyramid the forward lookup read
 *   coords, grad_out    as for alo_corr_lookup_backward
 *   grad_coords_levels  (B, num_levels, 2, H, W) float32
 */
int alo_corr_lookup_backward_coords(const float* const* levels, const float* coords, const float* grad_out,
                                    float* grad_coords_levels, int B, int H, int W, int radius, int num_levels, void* stream);

/*
 * ---- Extensions: one-pass epilogues of the layers that call the attention op -------------------------------------------
 * The reference evaluates these as separate PyTorch ops; they have no counterpart in its native code.  Both read and
 * write every byte once; all pointers 16-byte aligned; dtype ALO_F32 or ALO_BF16 (arithmetic in fp32 either way).
 *
 * alo_add_layernorm: out = LayerNorm_C(x + residual) * gamma + beta, and optionally out_pos = out + pos
 *   replaces  `src = self.norm1(src + self.dropout1(src2))` (+ the next layer's `with_pos_embed(src, pos)`) at inference
 *             alonet/deformable_detr/deformable_transformer.py:311-352,417-487
 *   x, residual (nullable), out, pos / out_pos (both or neither)   (rows, C) contiguous;  gamma, beta (C,)
 *   C % 4 == 0, C <= 1024.  `out` may alias `x` or `residual`.  Biased variance, eps inside the square root (torch.nn.LayerNorm).
 *
 * alo_bias_act: y = act(x + bias[c] [+ residual]),  act = ReLU when relu != 0, identity otherwise
 *   replaces  FrozenBatchNorm2d (folded into the convolution weights + this bias) -> [+ identity] -> ReLU of the ResNet
 *             bottleneck, alonet/detr/backbone.py:19-47 + torchvision Bottleneck.forward, on channels-last activations
 *   x, residual (nullable), y   (rows, C) contiguous = NHWC flattened;  bias (C,);  C % 4 == 0.  `y` may alias `x`.
 */
int alo_add_layernorm(const void* x, const void* residual, const void* gamma, const void* beta, void* out,
                      const void* pos, void* out_pos, long rows, int C, float eps, int dtype, void* stream);
int alo_bias_act(const void* x, const void* bias, const void* residual, void* y, long rows, int C, int relu,
                 int dtype, void* stream);

/*
 * alo_linear_shortk: y (M, N) = act(x (M, K) @ weight (N, K)^T + bias [+ residual (M, N)]), K in {64, 128, 256}, N % 64 == 0,
 * bf16 with fp32 accumulation, act = ReLU when relu != 0 (residual: the bottleneck's identity, added before the activation).  The short-K nn.Linear layers next to the op — value_proj, sampling_offsets,
 * attention_weights, output_proj of MSDeformAttn (ms_deform_attn.py:56-59), the FFN's first layer — and the backbone's 1x1
 * convolutions with K input channels over NHWC rows: memory-bound products; the weights stay in registers, x streams
 * through once per 256 output columns, y is written in whole lines (v_mfma_f32_32x32x16_bf16).  bias (N,) bf16 or NULL.
 * 16-byte aligned pointers.
 */
int alo_linear_shortk(const void* x, const void* weight, const void* bias, const void* residual, void* y, long M, int N,
                      int K, int relu, int dtype, void* stream);

/*
 * alo_value_proj_head_major: MSDeformAttn's value path in one kernel (ms_deform_attn.py:111-114): value_proj (a short-K
 * linear layer, as alo_linear_shortk), `masked_fill(input_padding_mask, 0)` and the head-major layout of
 * alo_value_head_major, all in the GEMM's epilogue.  x (batch * S, K) bf16, weight (heads * 32, K), bias (heads * 32,) or
 * NULL, padding_mask (batch * S,) uint8 or NULL -> value_hm (batch, heads, S, 32).  Head dimension 32, even head count.
 */
int alo_value_proj_head_major(const void* x, const void* weight, const void* bias, const void* padding_mask, void* value_hm,
                              int batch, int S, int heads, int K, int dtype, void* stream);

/*
 * alo_ffn256: y (M, 256) = relu(x (M, 256) @ w1 (F, 256)^T + b1) @ w2 (256, F)^T + b2, bf16 with fp32 accumulation,
 * F % 256 == 0: `linear2(relu(linear1(x)))` of the transformer layers (deformable_transformer.py:336-338,470-478) in one
 * kernel — the (M, F) hidden activation lives 64 rows at a time in LDS and never reaches memory.  b1 / b2 may be NULL.
 * w1 and w2 are PACKED weights: alo_pack_mfma_b(w (N, K) row-major) -> [N / 32][K / 16][64][8], the lane order of the MFMA
 * B operand, so that the weight stream is read in whole lines (pack once per weight update).
 */
int alo_pack_mfma_b(const void* w, void* packed, int N, int K, int dtype, void* stream);
int alo_ffn256(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* y, long M, int F,
               int dtype, void* stream);

/*
 * alo_linear_packed: y (M, N) = act(x (M, K) @ w (N, K)^T + bias [+ residual]) for the long-K 1x1 convolutions of the backbone and the
 * 1x1 input projections over NHWC rows (alonet/detr/backbone.py:19-47, deformable_detr.py:75-84): bf16, fp32 accumulation,
 * K % 256 == 0, N % 128 == 0.  w_packed = alo_pack_mfma_b(w).  residual (M, N) or NULL is added before the activation.
 */
int alo_linear_packed(const void* x, const void* w_packed, const void* bias, const void* residual, void* y, long M, int N, int K,
                      int relu, int dtype, void* stream);

/*
 * alo_conv1x1_nhwc: y (N, Ho, Wo, Cout) = act(conv1x1(x (N, H, W, Cin), stride) + bias [+ residual]) over NHWC maps: the strided
 * `downsample` convolutions of the bottlenecks (alonet/detr/backbone.py:84-92; torchvision Bottleneck) without a gathered copy of the
 * kept pixels — the tile loader of alo_linear_shortk (weight (Cout, Cin) row-major, weight_is_packed = 0, Cin in {64, 128, 256}) or
 * alo_linear_packed (weight = alo_pack_mfma_b(w), weight_is_packed = 1, Cin % 256 == 0, Cout % 128 == 0) addresses pixel
 * (n, stride * yo, stride * xo) itself.  Ho = (H - 1) / stride + 1.  residual (N, Ho, Wo, Cout) or NULL.
 */
int alo_conv1x1_nhwc(const void* x, const void* weight, int weight_is_packed, const void* bias, const void* residual, void* y, int N,
                     int H, int W, int Cin, int Cout, int stride, int relu, int dtype, void* stream);

/*
 * alo_conv3x3_small_nhwc: y (N, H, W, Cout) = conv3x3(x (N, H, W, Cin), stride 1, padding 1) + bias for FEW channels — Cin in
 * {16, 32, 64}, Cout = 1 or a multiple of 4 up to 32 — bf16 with fp32 accumulation: lay4 / lay5 / out_lay of PanopticHead's mask
 * decoder over B*Q maps (alonet/detr_panoptic/nn/FPNstyle.py:28-33,76-84).  w_frag = the weight in the MFMA row operand's fragment
 * order, [9 * Cin / 16 k-steps][64 lanes][8] bf16 with lane = 32 kg + m holding w[m][16 cs + 8 kg .. + 8][ky][kx] for k-step
 * (3 ky + kx) * (Cin / 16) + cs (rows m >= Cout are zero); bias32 = 32 fp32 values (zero-padded).
 */
int alo_conv3x3_small_nhwc(const void* x, const void* w_frag, const void* bias32, void* y, int N, int H, int W, int Cin, int Cout,
                           int dtype, void* stream);

/*
 * alo_conv3x3_nhwc: y (N, Ho, Wo, Cout) = act(conv3x3(x (N, H, W, Cin), stride 1 or 2, padding 1) + bias), bf16 with fp32
 * accumulation: Bottleneck.conv2 + the folded FrozenBatchNorm2d + ReLU of the ResNet backbone (alonet/detr/backbone.py:19-47,
 * 84-92; torchvision Bottleneck), an implicit GEMM on MFMA.  Ho = (H - 1) / stride + 1.  w_packed = alo_pack_mfma_b of the
 * (Cout, 9 * Cin) matrix w[o][(ky * 3 + kx) * Cin + c] (= the channels-last memory order of a (Cout, Cin, 3, 3) weight).
 * Cin % 64 == 0, Cout % 64 == 0; bias may be NULL.  When the output has too few tiles to fill the chip (the 2048 -> 256 input
 * projection, alonet/deformable_detr/deformable_detr.py:75-84) the reduction is split over the input channels: workspace is then
 * alo_conv3x3_workspace_bytes(...) bytes (0 = not needed; NULL = never split) of fp32 partial sums, added up in a fixed order.
 */
size_t alo_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride);
int alo_conv3x3_nhwc(const void* x, const void* w_packed, const void* bias, void* y, void* workspace, int N, int H, int W, int Cin,
                     int Cout, int stride, int relu, int dtype, void* stream);

/*
 * alo_stem_conv_pool: the ResNet stem in one kernel — y (N, Hp, Wp, 64) = maxpool3x3/s2/p1(relu(conv7x7/s2/p3(x) + bias)), bf16 with
 * fp32 accumulation (alonet/detr/backbone.py:19-47 with torchvision's ResNet.conv1 / bn1 / relu / maxpool; the frozen batch-norm
 * folded into weight and bias).  x is a (N, 3, H, W) bf16 image addressed through its ELEMENT strides (NCHW or channels-last);
 * Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1.  w_packed = alo_pack_mfma_b of the (64, 176) matrix
 * m[o][ky * 24 + kx * 3 + c] = w[o][c][ky][kx] (zero elsewhere).  The half-resolution convolution output never reaches memory.
 */
int alo_stem_conv_pool(const void* x, const void* w_packed, const void* bias, void* y, int N, int H, int W, long stride_n,
                       long stride_c, long stride_h, long stride_w, int dtype, void* stream);

/*
 * alo_groupnorm_rows: GroupNorm of channels-last rows x (B, HW, C) -> y (B, HW, C) with an arbitrary batch stride, so that the
 * result lands inside the encoder's flattened (B, S, C) source at the level's offset: `input_proj[l][1]` + the flatten / transpose /
 * cat of DeformableTransformer.forward (alonet/deformable_detr/deformable_detr.py:75-76,141-151; deformable_transformer.py:331-337).
 * bf16 in / out, fp32 Welford statistics (biased variance, as torch.nn.GroupNorm), deterministic.  weight / bias (C,) bf16.
 * C / 8 and groups must divide 256; C / groups % 8 == 0.  workspace: alo_groupnorm_rows_workspace_bytes(B, HW, groups) bytes.
 */
size_t alo_groupnorm_rows_workspace_bytes(int B, int HW, int groups);
int alo_groupnorm_rows(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW, int C,
                       int groups, float eps, long y_batch_stride, int dtype, void* stream);
/*
 * alo_groupnorm_rows_act: the same with 2 or 4 channels per group as well (a thread's 8 channels then span several groups) and an
 * optional ReLU in the normalise pass: the GroupNorm(8, 32) / GroupNorm(8, 16) + ReLU of PanopticHead's mask decoder over B*Q maps of
 * up to 200 x 334 pixels (alonet/detr_panoptic/nn/FPNstyle.py:24-36,60-84), channels-last in and out — ATen's GroupNorm takes NCHW,
 * i.e. a layout copy either side of every one of them.
 */
int alo_groupnorm_rows_act(const void* x, const void* weight, const void* bias, void* y, void* workspace, int B, int HW, int C,
                           int groups, float eps, long y_batch_stride, int relu, int dtype, void* stream);

/*
 * alo_upsample_add_nhwc: out (BQ, H, W, C) = fpn[bq / Q] (B, H, W, C) + nearest-up-sampled x_low (BQ, h, w, C), channels-last bf16:
 * `_expand(adapter(fpn), Q) + F.interpolate(x, size=(H, W), mode="nearest")` of the mask decoder's FPN steps (FPNstyle.py:60-84) in
 * one pass — without the per-query copy of the adapter output, the up-sampled copy and the add.  Index arithmetic of ATen's nearest
 * kernel (scale = float(in) / out, src = min(int(floorf(dst * scale)), in - 1)); fp32 add, one rounding: bit-identical to the stock ops.
 */
int alo_upsample_add_nhwc(const void* x_low, const void* fpn, void* out, int BQ, int Q, int C, int h, int w, int H, int W, int dtype,
                          void* stream);

/*
 * alo_mask_pyramid: the padding mask of every level, flattened, and the valid ratios, from the (B, H, W) frame mask:
 *   mask_flat[b][start_l + y * w_l + x] = F.interpolate(mask.float(), (h_l, w_l), mode_l)[b, 0, y, x] != 0   (uint8 0 / 1)
 *   valid_ratios[b][l] = (un-padded pixels in the first row / w_l, in the first column / h_l)                    (fp32)
 * mode_l = "nearest" where bit l of nearest_levels is set, else "bilinear", align_corners = False, in ATen's arithmetic.  Replaces
 * alonet/detr/backbone.py:127-128 (mask resize per stage), deformable_detr.py:147, deformable_transformer.py:318-323,334-338,350.
 * frame_mask: fp32 (mask_is_float = 1) or uint8 / bool, non-zero = padding.  level_shapes_host: 2 * L ints (h_0, w_0, ...) in HOST memory.
 *
 * alo_encoder_reference_points: (B, S, L, 2) fp32 — the centre of every pixel of every level normalised by the valid extent of its own
 * level and re-scaled by each level's valid ratio (DeformableTransformerEncoder.get_reference_points, deformable_transformer.py:136-149).
 */
int alo_mask_pyramid(const void* frame_mask, int mask_is_float, unsigned char* mask_flat, float* valid_ratios, int B, int H, int W,
                     int L, const int* level_shapes_host, unsigned nearest_levels, void* stream);
int alo_encoder_reference_points(const float* valid_ratios, float* reference_points, int B, int L, const int* level_shapes_host,
                                 void* stream);

/*
 * alo_panoptic_onehot: PanopticHead.inference's mask post-processing in one pass (alonet/detr_panoptic/detr_panoptic.py:96-110):
 * onehot[b][q][y][x] = 1 where q is the arg-max over the image's queries of threshold(sigmoid(bilinear(mask_logits[b][q], (H, W))))
 * (F.threshold(., threshold, 0); lowest query on ties) and at least one query passed the threshold, else 0.  mask_logits (B, Q, h, w)
 * fp32, onehot (B, Q, H, W) int64 (aloscene.Mask's dtype).  Bilinear arithmetic as ATen's upsample_bilinear2d, align_corners = False.
 */
int alo_panoptic_onehot(const float* mask_logits, long long* onehot, int B, int Q, int h, int w, int H, int W, float threshold,
                        void* stream);

/*
 * alo_pos_sine_flat: the sine positional encoding of every level of the pyramid, written straight into the flattened
 * (B, S, 2F) layout the encoder consumes, level embedding added: what PositionEmbeddingSine.forward + the
 * `pos.flatten(2).transpose(1, 2) + level_embed[lvl]` / cat of DeformableTransformer.forward compute with ~15 PyTorch
 * kernels per level (alonet/transformers/position_encoding.py:29-72, deformable_transformer.py:562-577).
 *   padding_mask (B, S) uint8/bool (1 on padding), dim_t (F,) fp32 = temperature ** (2 * (i // 2) / F),
 *   level_embed (L, 2F) of `dtype` or NULL, out (B, S, 2F) of `dtype`, workspace B * S * 2 floats.  F % 4 == 0.
 */
int alo_pos_sine_flat(const void* padding_mask, const int32_t* spatial_shapes, const int32_t* level_start_index,
                      const float* dim_t, const void* level_embed, void* out, float* workspace, int B, int S, int L,
                      int num_pos_feats, int normalize, int center, float scale, float eps, int dtype, void* stream);

/*
 * ---- Extensions: elementwise glue of RAFT's update block (alonet/raft/update.py:27-33,83-101), fp32, NCHW ------------------
 * One pass each; H*W % 4 == 0; pointers 16-byte aligned.  The convolutions stay on MIOpen and are called WITHOUT bias.
 *
 * alo_bias_act_nchw: y[b,c,:] = act(x[b,c,:] + bias[c])            (convolution bias + ReLU; y may alias x)
 * alo_gru_gate:      zr (B, 2C, H, W) = pre-activations [z | r] of one convolution over [h | x]
 *                    z  <- sigmoid(z + bias_zr[:C])                 written back over its own slot in zr
 *                    rh <- sigmoid(r + bias_zr[C:]) * h             written into the first C channels of the [r*h | x] buffer
 * alo_gru_update:    h  <- (1 - z) * h + z * tanh(q + bias_q)       in place in the [h | x] buffer (+ contiguous copy `net`)
 *   h / rh are channel slices of (B, C + Cx, H, W) buffers: *_batch_stride = (C + Cx) * H * W elements.
 */
int alo_bias_act_nchw(const float* x, const float* bias, float* y, int B, int C, int HW, int relu, void* stream);
int alo_gru_gate(float* zr, const float* bias_zr, const float* h, float* rh, int B, int C, int HW,
                 long h_batch_stride, long rh_batch_stride, void* stream);
int alo_gru_update(const float* q, const float* bias_q, const float* zr, float* h, float* net, int B, int C, int HW,
                   long h_batch_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALO_HOTPATH_H */
