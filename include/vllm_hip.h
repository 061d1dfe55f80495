/*
 * vllm_hip.h -- C ABI of libvllm_hip.so: the MI355X (gfx950) image->visual-token hot path of VisionLLMv2.
 *
 * Plain pointers and sizes only (no torch / ATen types).  Every pointer marked "device" is a HIP device
 * pointer owned by the caller (PyTorch's caching allocator in practice); the library never allocates or
 * frees device memory and never synchronises: all work is enqueued on `stream` (a hipStream_t passed as
 * void*; NULL = the legacy default stream).  Thread-safety: entry points are re-entrant.  Global state: the per-thread
 * last-error string, and the PROCESS-WIDE tuning knobs of vllm_set_option() (plain ints read at launch time: change them
 * only while no other thread is launching) plus one-time per-process caches (device CU count, kernel attributes).
 *
 * Return value: 0 on success, negative VLLM_E* on error (vllm_last_error() gives the message).  The Python
 * mirror (visionllm_amd/_lib.py) turns non-zero into RuntimeError, as the reference's C++ exceptions do
 * (AT_ASSERTM in visionllmv2/model/unipose/ops/src/cuda/ms_deform_attn_cuda.cu:28-52).
 *
 * Reference paths are relative to /root/reference/VisionLLMv2/.
 */
#ifndef VLLM_HIP_H
#define VLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout or an entry point's meaning changes.  2 (round 4): VllmVitLayer grew by six pointers
 * (an ARRAY of these is passed, so the stride changed) and VllmMsdaLayerDesc got `geometry` / `reserved0` in front of its
 * pointers -- a caller built against version 1 must not run against this library: check vllm_abi_version() == VLLM_ABI_VERSION
 * at load time and the vllm_*_sizeof() of every descriptor it fills. */
#define VLLM_ABI_VERSION 2

#define VLLM_OK 0
#define VLLM_EINVAL (-1)   /* bad argument (shape / alignment / unsupported size) */
#define VLLM_ELAUNCH (-2)  /* hipLaunchKernel / HIP runtime error */
#define VLLM_ENOTIMPL (-3)

typedef void *vllm_stream_t; /* hipStream_t */

int vllm_abi_version(void);
const char *vllm_last_error(void);
/* Fills name[0..cap) with the device's gcnArchName; returns CU count or negative error. */
int vllm_device_info(char *name, int cap);
/* Tuning / test knobs (process-wide).  "msda_tiled": encoder-shaped (Lq == S) MSDA forward kernel, same results to fp32
 * rounding: 0 plain gather kernel; 1 automatic (default): generation 9 since the end of round 4 (msda_tiled9.hip: pyramid items, two
 * teams of waves half a period apart, software-pipelined gather; round 3: generation 8, round 2: generation 7)
 * when the level maps are nested halves -- decided on the device, no host sync -- else generation 4;
 * 2 generation 4 with 8 waves per block; 3 generation 2; 5 generation 4 with the phase clock (vllm_debug_counters);
 * 8 generation 4, 560-pixel windows, 2 blocks per CU; 9 generation 4, 360 pixels, 3 blocks per CU (the round-1 default);
 * 10-14 generation 6 (msda_tiled6.hip; 10 / 14 with phase clock, 11-13 gather / staging variants); 17 generation 6; 18 generation 8
 * (msda_tiled8.hip, what "automatic" ran in round 3), 19 generation 8 with the phase clock, 20 generation 9 (= automatic), 21 generation 9
 * with the phase clock.  15 / 16
 * (generation 7) are rejected since round 4: that kernel is tools/experiments/msda_tiled7.hip.  "gemm_variant": 0 auto, 1 128x128 kernel, 2 256x256 8-phase
 * kernel, 4 8-phase kernel on the 32x32x16 MFMA.  "gemm_direct_store": the 8-phase kernel's epilogue goes 0 through LDS
 * (row-contiguous 16-byte stores), 1 straight from the accumulator layout, 2 automatic (default; same results either way).
 * "attn_variant": 32 automatic (default) = 2 | 64 with the class-token split.  bit1 deferred rescale, bit4 do not trim padding keys /
 * padding query waves (and no class-token split), bit6 O leaves through LDS as whole rows, bit7 NO class-token split (S = 64 n + 1:
 * token 0 as the initial softmax state of every query, and -- where the n^2 body rows leave a spare wave in their last block -- as
 * that wave's only query row; round 6), bit10 token 0 out of the key tiling only.  Bits 0, 2, 3 selected rounds 2-5 schedules that
 * never became the default (software-pipelined K, s_setprio, hoisted transpose reads); they are ignored since round 6.
 * "dcnv3_tiled": DCNv3 forward for fp32, group channels 16 / 32, <= 9 points: 1 (default) the pipelined LDS-tiled kernel
 * (dcnv3_pipe.hip), 3 the two-blocks-per-CU LDS-tiled kernel (dcnv3_tiled.hip), 0 the gather kernel (same results to fp32
 * rounding); 2 / 4 = 1 / 3 with the phase clock (vllm_debug_counters then reads IT).
 * "msda_layer_fused": 1 (default) vllm_msda_layer_forward runs sampling_offsets + attention_weights as one GEMM whose
 * epilogue does the softmax and the location arithmetic, and takes the operator's result in bf16 straight from the
 * LDS-tiled kernel (needs L * P == 16, P even; other layers compose automatically); 0 the explicit composition (two GEMMs,
 * prep kernel, fp32 operator, conversion pass) -- the A/B reference, same results to fp32 rounding.
 * "gemm_tile_rb" (round 4; VLLM_GEMM_TILE_RB): tile order of the persistent 8-phase GEMM: -1 automatic (default: bands of 4 row
 * panels when the weight has >= 32 column tiles, 8 when >= 8, else the dense order), 0 dense, RB > 0 bands of RB row panels per
 * XCD rectangle; every order gives the same bits.  "gemm_skinny" (round 4; VLLM_GEMM_SKINNY): 1 (default) K = 256, N = 256 / 256 + 128
 * GEMMs with >= 4096 rows (the linears of a d_model = 256 deformable-attention layer) run on the weight-stationary streaming kernel
 * (gemm_skinny.hip); 0 on the 128 x 128 tile kernel -- the same bits either way.
 * "dcnv3_bwd_tiled" (round 4; VLLM_DCNV3_BWD_TILED): 1 (default) vllm_dcnv3_backward_f32 with group channels 32 runs on the windowed
 * kernel (grad_input as S^T x grad_out on the fp32 MFMA: msda_bwd_mfma.hip with the DCN flag), 0 on the gather kernel (global atomics
 * per (point, corner, channel)); the same gradients to fp32 rounding.
 * "gemm_half_tail" (round 4; VLLM_GEMM_HALF_TAIL): 1 (default) the persistent 8-phase GEMM runs the tiles of its last, incomplete round
 * as two half-height tiles each when they then still fit the grid (qkv at 40 ViT-L tiles: 4.27 rounds of work in 5 -> 4 + a half-tile
 * round); 0 whole tiles.  The same bits either way.
 * "msda_layer_value_bf16" (round 4; VLLM_MSDA_LAYER_VALUE_BF16): 1 (default) vllm_msda_layer_forward stores the projected value in
 * bf16 when the query set is not the value pyramid (decoder cross-attention: the gather kernel reads bf16 natively; the reference's
 * bf16 module rounds the value to bf16 too) and the streaming value GEMM serves the shape; 0 fp32 value everywhere.
 * Environment variables VLLM_MSDA_TILED / VLLM_GEMM_VARIANT / VLLM_ATTN_VARIANT give the initial values.  Returns the
 * previous value or VLLM_EINVAL for an unknown name / value.  (Measured-slower experiments -- MSDA generations 3 and 5, the
 * 4-wave GEMM, the two-row-group attention kernel -- live under tools/experiments/ and are not part of the library.) */
int vllm_set_option(const char *name, int value);
/* Diagnostics.  Process-wide and NOT for production use: VLLM_GEMM_PROF=1 reroutes vllm_debug_counters to the GEMM for the
 * whole process; VLLM_GEMM_TRACE (a raw device address the 8-phase GEMM writes per-block timestamps to) exists only in
 * builds with -DVLLM_GEMM_TRACE_ENABLE, where the address is validated as device memory first.
 * (VLLM_GEMM_PROF=1 in the environment: the 8-phase GEMM's prologue / main loop / epilogue ticks + block count;
 * "dcnv3_tiled" = 2: the DCNv3 kernel's phases; otherwise:) with "msda_tiled" = 5 / 10 / 14 / 16 the LDS-tiled MSDA kernel of that generation adds per-phase
 * shader-clock ticks to 16 device counters; this reads the current generation's into out[0..n) and clears them.  Returns the
 * number of counters written. */
int vllm_debug_counters(long *out, int n);
/* In-step kernel timing (measurement; process-wide, off by default).  vllm_prof_enable(1) starts a recording: every
 * operator the library enqueues is preceded by a HIP event on its stream; vllm_prof_read waits for the last one and returns,
 * per tag (vllm_prof_tag_name), the summed time in microseconds from each mark to the next one and the number of marks --
 * a kernel's duration INSIDE the step, queueing and launch gaps included (bench.py's `us_per_launch`).  Both return the
 * number of tags (read: slots needed); vllm_prof_enable(0) stops and discards. */
int vllm_prof_enable(int on);
int vllm_prof_read(double *us_sum, long *count, int n);
const char *vllm_prof_tag_name(int tag);

/* ------------------------------------------------------------------------------------------------
 * B3. Multi-scale deformable attention (MSDA) operator.
 *
 * Replaces:  ms_deform_attn_forward / ms_deform_attn_backward
 *   visionllmv2/model/unipose/ops/src/ms_deform_attn.h:20-61          (pybind module MultiScaleDeformableAttention,
 *   visionllmv2/model/unipose/ops/src/vision.cpp:13-16)
 *   visionllmv2/model/unipose/ops/src/cuda/ms_deform_attn_cuda.cu:20-80, 83-153  (host launchers)
 *   visionllmv2/model/unipose/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-298     (forward kernel)
 *   mmcv/mmcv/ops/csrc/pytorch/ms_deform_attn.cpp:38-60               (mmcv._ext twins)
 *
 * Layouts (all contiguous, as the reference asserts):
 *   value   [B, S, M, D]          S = sum_l H_l*W_l
 *   shapes  [L, 2] int64 (H, W)   DEVICE memory (read by the kernel with scalar loads, like the reference)
 *   lsi     [L]    int64          DEVICE memory, level start index
 *   loc     [B, Lq, M, L, P, 2]   (x, y) in [0,1] (out-of-range allowed -> zero padding)
 *   attw    [B, Lq, M, L, P]
 *   out     [B, Lq, M*D]          fully overwritten (no pre-zeroing needed)
 * `im2col_step` of the reference only chunks the batch over several launches; here it is one launch and the
 * argument does not exist (the Python mirror accepts and checks it: B % min(B, im2col_step) == 0).
 * ------------------------------------------------------------------------------------------------ */
int vllm_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attw,
                          int B, int S, int M, int D, int L, int Lq, int P,
                          float *out, vllm_stream_t stream);
int vllm_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                          const double *loc, const double *attw,
                          int B, int S, int M, int D, int L, int Lq, int P,
                          double *out, vllm_stream_t stream);
/* vllm_msda_forward_f32 for a caller that knows the level geometry on the HOST.  `shapes` is device memory (as in the
 * reference), so by itself the library cannot know whether the level maps are nested halves whose cells are the Lq
 * queries (the det heads' 168^2 / 84^2 / 42^2 / 21^2 encoder case, or the ceil-divided 100x167 / 50x84 / ... of a detection
 * backbone: served by the pyramid-item kernel) without a host synchronisation: VLLM_GEO_UNKNOWN enqueues the pyramid
 * kernel's two instantiations AND the any-geometry kernel and the device picks (two empty launches).  The reference's modules synchronise once per forward pass anyway (`(H * W).sum() == Len_in`,
 * ms_deform_attn.py:100); the Python mirror learns the geometry in that same read-back and passes it here: exactly one
 * launch.  A PYRAMID / NESTED hint that the device-side test contradicts traps (the hint is never trusted for addressing). */
#define VLLM_GEO_UNKNOWN 0
#define VLLM_GEO_PYRAMID 1
#define VLLM_GEO_GENERAL 2
#define VLLM_GEO_NESTED 3   /* 1-4 levels, each the previous one halved, rounded either way (ceil- / floor-divided maps), not all exact */
int vllm_msda_forward_f32_geo(const float *value, const int64_t *shapes, const int64_t *lsi,
                              const float *loc, const float *attw,
                              int B, int S, int M, int D, int L, int Lq, int P, int geometry,
                              float *out, vllm_stream_t stream);
/* bf16 value/out, fp32 loc/attw, fp32 accumulation (extension: the reference upcasts bf16 to fp32 first,
 * modeling_ov_grounding_dino_mask_dn.py:764-766; this variant halves the gathered bytes). */
int vllm_msda_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *attw,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           uint16_t *out, vllm_stream_t stream);
/* Integer part of the sampling, for index-exact parity tests: per point (b,q,m,l,p)
 * h_low, w_low (int32) and mask (bit0 accepted, bits1..4 corners 1..4 in bounds), computed by the same
 * device function the forward kernels use. */
int vllm_msda_sample_index_f32(const int64_t *shapes, const float *loc,
                               int B, int M, int L, int Lq, int P,
                               int32_t *h_low, int32_t *w_low, uint8_t *mask, vllm_stream_t stream);
/* Backward (B3, row f1 of SURVEY.md section 8).  grad_* must be zero-filled by the caller, exactly as the
 * reference does with at::zeros (ms_deform_attn_cuda.cu:118-120; mmcv multi_scale_deform_attn.py:80-94). */
int vllm_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *attw, const float *grad_out,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           float *grad_value, float *grad_loc, float *grad_attw, vllm_stream_t stream);
/* 1 when vllm_msda_backward_f32 with these arguments writes EVERY element of grad_loc and grad_attw itself (the matrix-core kernel of
 * the encoder self-attention shape: the gradients of a rejected point are stored as zeros), so the caller may pass those two
 * uninitialised -- 460 MB of memset less at BASELINE cfg 4, B = 8; grad_value must be zero-filled in every case.  0: zero-fill all three. */
int vllm_msda_backward_f32_writes_point_grads(const float *value, const float *loc, const float *grad_out, const float *grad_value,
                                              const float *grad_loc, int B, int S, int M, int D, int L, int Lq, int P);
int vllm_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                           const double *loc, const double *attw, const double *grad_out,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           double *grad_value, double *grad_loc, double *grad_attw, vllm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a13 / f2. The whole deformable-attention LAYER around the operator, for bf16 (inference-dtype) modules.
 *
 * Replaces the bodies of
 *   MSDeformAttn.forward                                   visionllmv2/model/unipose/ops/modules/ms_deform_attn.py:83-145
 *   MultiScaleDeformableAttention.forward (mmcv)           mmcv/ops/multi_scale_deform_attn.py:262-367
 *   GroundingDinoMultiscaleDeformableAttention.forward     visionllmv2/model/grounding_dino/modeling_grounding_dino_mask_dn.py:706-784
 * i.e. value_proj (+ key-padding zero fill), the sampling_offsets / attention_weights linears, softmax over the
 * L*P logits of a head, the location arithmetic (2-d reference points: ref + off / (W_l, H_l); 4-d: ref_xy +
 * off / P * ref_wh * 0.5, or the UniPose "4D normalizer" form off / (W_l, H_l) * ref_wh * 0.5), the operator and
 * output_proj.  Activations and weights are bf16 (nn.Linear layout [out, in]); every INTERNAL tensor (value, offsets,
 * logits, locations, weights, operator output) is kept in fp32, i.e. at or above the precision the reference has
 * when it upcasts around the operator (ms_deform_attn.py:131-139).
 * ------------------------------------------------------------------------------------------------ */
typedef struct VllmMsdaLayerDesc {
    int32_t d_model, n_heads, n_levels, n_points;
    int32_t ref_dim;             /* last dim of reference_points: 2 or 4 */
    int32_t use_4d_normalizer;   /* ref_dim 4 only: UniPose's use_4D_normalizer */
    int32_t geometry;            /* VLLM_GEO_*: what the host knows about spatial_shapes (0 = nothing) */
    int32_t reserved0;           /* 0 */
    const uint16_t *value_proj_w, *value_proj_b;                  /* [C, C], [C]           (device, bf16) */
    const uint16_t *sampling_offsets_w, *sampling_offsets_b;      /* [M*L*P*2, C], [M*L*P*2] */
    const uint16_t *attention_weights_w, *attention_weights_b;    /* [M*L*P, C], [M*L*P] */
    const uint16_t *output_proj_w, *output_proj_b;                /* [C, C], [C] */
} VllmMsdaLayerDesc;
unsigned long vllm_msda_layer_desc_sizeof(void);
/* Bytes of device scratch vllm_msda_layer_forward needs for (B, Lq, S); negative on a bad descriptor. */
long vllm_msda_layer_workspace_bytes(const VllmMsdaLayerDesc *desc, int B, int Lq, int S);
/* query [B, Lq, C] bf16 (position embedding already added by the caller), reference_points [B, Lq, L, ref_dim] fp32,
 * input_flatten [B, S, C] bf16, padding_mask [B, S] uint8 (non-zero = padded) or NULL, spatial_shapes [L, 2] /
 * level_start_index [L] device int64 (H, W), out [B, Lq, C] bf16.  d_model % 64 == 0, d_model / n_heads % 4 == 0. */
int vllm_msda_layer_forward(const VllmMsdaLayerDesc *desc, const uint16_t *query, const float *reference_points,
                            const uint16_t *input_flatten, const uint8_t *padding_mask, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, int B, int Lq, int S, uint16_t *out, void *workspace,
                            long workspace_bytes, vllm_stream_t stream);
/* The two elementwise kernels of the layer, exposed for the parity tests:
 * in place, offsets [R, M, L, P, 2] -> sampling locations and logits [R, M, L*P] -> softmax weights (R = B*Lq rows,
 * reference_points [R, L, ref_dim]); and fp32 -> bf16 (round to nearest even) of n elements. */
int vllm_msda_prep_f32(float *offsets_to_locations, float *logits_to_weights, const float *reference_points,
                       const int64_t *spatial_shapes, long R, int M, int L, int P, int ref_dim, int use_4d_normalizer,
                       vllm_stream_t stream);
int vllm_f32_to_bf16(const float *src, uint16_t *dst, long n, vllm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f3. DCNv3 forward (InternImage det backbone).
 *
 * Replaces:  dcnv3_forward   visionllmv2/model/ops_dcnv3/src/dcnv3.h:20-39, src/cuda/dcnv3_cuda.cu:20-80
 *            (kernel src/cuda/dcnv3_im2col_cuda.cuh:217-278; Python caller functions/dcnv3_func.py:39-43).
 * input [N, H, W, G*C], offset [N, Ho, Wo, G*kh*kw*2] (x, y pairs, kernel_w-outer / kernel_h-inner point order),
 * mask [N, Ho, Wo, G*kh*kw], out [N, Ho, Wo, G*C]; Ho = (H + 2*ph - (dh*(kh-1)+1)) / sh + 1 (likewise Wo).  All device,
 * contiguous.  im2col_step of the reference is a batching detail of its launch loop and has no counterpart (one launch).
 * ------------------------------------------------------------------------------------------------ */
int vllm_dcnv3_forward_f32(const float *input, const float *offset, const float *mask, int N, int H, int W, int G, int C,
                           int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale, float *out,
                           vllm_stream_t stream);
int vllm_dcnv3_forward_f64(const double *input, const double *offset, const double *mask, int N, int H, int W, int G, int C,
                           int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, double offset_scale, double *out,
                           vllm_stream_t stream);
/* Half precision (round 5): the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (dcnv3_cuda.cu:69) with opmath_t = float.
 * IEEE binary16 tensors as uint16_t bit patterns; fp32 arithmetic, the output rounded to nearest even once. */
int vllm_dcnv3_forward_f16(const uint16_t *input, const uint16_t *offset, const uint16_t *mask, int N, int H, int W, int G, int C,
                           int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale, uint16_t *out,
                           vllm_stream_t stream);
/* Backward (round 4).  Replaces DCNv3.dcnv3_backward (ops_dcnv3/src/dcnv3.h:41-64, cuda/dcnv3_cuda.cu:92-174; kernels
 * dcnv3_im2col_cuda.cuh:86-146, 279-857), called by DCNv3Function.backward (functions/dcnv3_func.py:51-59).
 * grad_output [N, Ho, Wo, G*C]; grad_input [N, H, W, G*C] MUST BE ZERO-FILLED by the caller (the reference's host code allocates it
 * with at::zeros_like; the sums arrive by floating-point atomics); grad_offset [N, Ho, Wo, G*kh*kw*2] and grad_mask
 * [N, Ho, Wo, G*kh*kw] are written completely.  Any channel count (group channels 4 / 8 / 16 / 32 / 64: the vectorised kernel). */
int vllm_dcnv3_backward_f32(const float *input, const float *offset, const float *mask, const float *grad_output, int N, int H, int W,
                            int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale,
                            float *grad_input, float *grad_offset, float *grad_mask, vllm_stream_t stream);
int vllm_dcnv3_backward_f64(const double *input, const double *offset, const double *mask, const double *grad_output, int N, int H, int W,
                            int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, double offset_scale,
                            double *grad_input, double *grad_offset, double *grad_mask, vllm_stream_t stream);

/* Half-precision backward (round 5; dcnv3_cuda.cu:147 dispatches AND_HALF too): operands and gradients are binary16, the arithmetic
 * is the fp32 backward on widened copies in the CALLER's workspace (vllm_dcnv3_backward_f16_workspace bytes, 16-byte aligned; -1 for
 * an invalid geometry), every gradient rounded once.  grad_input need not be zero-filled here (the fp32 accumulator is). */
long vllm_dcnv3_backward_f16_workspace(int N, int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);
int vllm_dcnv3_backward_f16(const uint16_t *input, const uint16_t *offset, const uint16_t *mask, const uint16_t *grad_output, int N, int H,
                            int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale,
                            uint16_t *grad_input, uint16_t *grad_offset, uint16_t *grad_mask, void *workspace, long workspace_bytes,
                            vllm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f4. Region-encoder point sampling.
 *
 * Replaces: point_sample (F.grid_sample(input, 2*coords-1), bilinear, zeros padding, align_corners=False) and the masked
 * mean over a region's sampled points, visionllmv2/model/region_encoder.py:24-47, 127-141.
 * input [N, C, H, W] fp32 (contiguous NCHW), coords [N, P, 2] (x, y) in [0, 1], valid [N, P] uint8.
 * ------------------------------------------------------------------------------------------------ */
int vllm_point_sample_f32(const float *input, const float *coords, int N, int C, int H, int W, int P, float *out /* [N,C,P] */,
                          vllm_stream_t stream);
/* out[n, c] = sum_p valid * sample / sum_p valid, 0 for a region without points ((x / 0).nan_to_num() of the reference). */
int vllm_point_sample_mean_f32(const float *input, const float *coords, const uint8_t *valid, int N, int C, int H, int W, int P,
                               float *out /* [N,C] */, vllm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Building blocks of the ViT path (bf16 storage, fp32 accumulation).  Exposed individually so the parity
 * tests can pin every kernel against the oracle, and as bring-up hooks B4/B5 of SURVEY.md section 8b.
 * All bf16 tensors are passed as uint16_t*.
 * ------------------------------------------------------------------------------------------------ */

/* Epilogues of vllm_gemm_bf16 */
#define VLLM_EPI_BIAS 0        /* y = x W^T + b                                   (nn.Linear) */
#define VLLM_EPI_GELU 1        /* y = gelu_erf(x W^T + b)                         (InternMLP.fc1+act, vl_bridge GELU) */
#define VLLM_EPI_QUICK_GELU 2  /* y = z*sigmoid(1.702 z)                          (CLIP MLP) */
#define VLLM_EPI_RESIDUAL 3    /* y = res + (x W^T + b) * scale                   (LayerScale + residual, modeling_intern_vit.py:206-208) */
#define VLLM_EPI_EMBED 4       /* patch embedding: rows scattered past the CLS slot, + position embedding */
#define VLLM_EPI_F32 5         /* y = x W^T + b kept in fp32: Y is float*, ldy in floats (16-byte aligned rows); `res`, if
                                * given, is a uint8 row mask [M]: masked rows are written as zeros (the key-padding
                                * zero-fill of the MSDA value projection) */
/* Kernel choice is automatic (256x256 8-phase schedule for M,N >= 1024, 128x128 otherwise); OR one of these into
 * `epilogue` to force a schedule (parity tests / tuning only). */
#define VLLM_GEMM_FORCE_128 0x100
#define VLLM_GEMM_FORCE_256 0x200   /* 8-phase schedule, 256-row block tile */
#define VLLM_GEMM_FORCE_192 0x300   /* 8-phase schedule, 192-row block tile */
#define VLLM_GEMM_FORCE_4W 0x400    /* 4-wave schedule: 256x256x32 block tile, 128x128 per wave */
#define VLLM_GEMM_FORCE_MF32 0x800  /* 8-phase schedule, 256-row block tile, v_mfma_f32_32x32x16_bf16 */
#define VLLM_GEMM_FORCE_TILEWISE 0x1000 /* 8-phase schedule with one workgroup per tile, where the persistent walk over the tiles
                                        * (qkv / fc1 shapes: bias / GELU / quick-GELU epilogue, N % 256 == 0) would be taken */

/* Y[M,N] = epilogue(X[M,K] @ W[N,K]^T + bias).  Replaces F.linear / nn.Conv2d-as-GEMM on the path
 * (modeling_intern_vit.py:112,124,128,141,172-178; modeling_visionllmv2.py:162-182).
 * K % 64 == 0, N % 4 == 0, 16-byte aligned operands.  bias/scale may be NULL.  For VLLM_EPI_EMBED `res` is the
 * position table [P+1, N] (row stride ldr) and P the patches per image; output row of input row m is
 * (m / P) * (P + 1) + 1 + m % P. */
int vllm_gemm_bf16(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y,
                   int M, int N, int K, int ldx, int ldw, int ldy, int epilogue,
                   const uint16_t *scale, const uint16_t *res, int ldr, int P, vllm_stream_t stream);

/* The same GEMM with caller-provided scratch for the stream-K tail of the 8-phase schedule: when the tiles of a GEMM do not
 * fill a whole number of rounds on the device's CUs, the K iterations of the last round's tiles are spread evenly over all CUs
 * (fp32 partial tiles meet in `scratch`, summed in a fixed order: results are run-to-run identical).  `scratch` is
 * vllm_gemm_scratch_bytes() bytes, 16-byte aligned; the call resets its first 4096 bytes (flags) with a memset node in front of
 * the kernel; calls sharing a scratch must be ordered on one stream.  NULL scratch = vllm_gemm_bf16.  vllm_vit_forward /
 * vllm_bridge_forward reserve theirs inside their workspace. */
long vllm_gemm_scratch_bytes(void);
long vllm_gemm_sk_launches(void);   /* GEMM launches of this process that took the stream-K tail (tests / tuning) */
long vllm_gemm_half_tail_launches(void);   /* ... of the persistent schedule whose last round ran as half-height tiles (round 4) */
long vllm_gemm_persistent_launches(void);   /* GEMM launches of this process that took the persistent 8-phase schedule (tests / tuning) */
int vllm_gemm_bf16_sk(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y,
                      int M, int N, int K, int ldx, int ldw, int ldy, int epilogue,
                      const uint16_t *scale, const uint16_t *res, int ldr, int P,
                      void *scratch, long scratch_bytes, vllm_stream_t stream);

/* The same GEMM with a LayerNorm / RMSNorm folded into it (8-phase schedule only; what vllm_vit_forward does between the
 * residual GEMMs and the qkv / fc1 GEMMs instead of launching the norm, modeling_intern_vit.py:198-210 / CLIPEncoderLayer):
 *   producer (ln_out != NULL): besides Y it writes, per output row and 256-column tile, {mean, M2 = sum (y - mean)^2} of the
 *     bf16 values it stored ({sum y^2, 0} with ln_rms) to ln_out [M][ceil(N / 256)][2] (fp32);
 *   consumer (ln_in != NULL): X holds the UN-normalised rows (K elements each, 768 < K <= 1024: ln_slots == 4), ln_in their
 *     statistics [M][4][2] as a producer with N == K leaves them, W the weight with the norm's gamma multiplied in (W'[n,k] = gamma_k W[n,k], rounded to
 *     bf16), ln_colsum[n] = sum_k W'[n,k] (fp32; LayerNorm only), ln_bias[n] = b_n + sum_k beta_k W[n,k] (fp32, may be NULL; it
 *     replaces `bias`):  y = epilogue(r_m * (x W'^T) - r_m mean_m colsum_n + ln_bias_n),  r_m = rsqrt(var_m + ln_eps).
 * The row statistics are exact fp32 (the reference rounds the normalised tensor to bf16 first: the folded form differs from it
 * by that rounding, i.e. it is the more accurate of the two).
 * Round 5, WIDE statistics (InternViT-6B, hidden 3200 = 12.5 column tiles): with ln_rms and ln_slots != 4 (1 .. 16; producer:
 * ceil(N / 256), consumer: ceil(K / 256), N and K multiples of 8) the buffer is [M][16] floats, slot s = sum y^2 over column tile s,
 * slots >= ln_slots ZERO (the caller clears the buffer once; the producer never writes them); the consumer adds all 16 in a fixed
 * order.  Only the persistent schedule implements it (M, N large enough for at least one tile per CU): a call it does not take
 * returns VLLM_EINVAL and the caller launches the norm instead (vllm_vit_forward asks with a dry run first). */
int vllm_gemm_bf16_ln(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y,
                      int M, int N, int K, int ldx, int ldw, int ldy, int epilogue,
                      const uint16_t *scale, const uint16_t *res, int ldr,
                      float *ln_out, const float *ln_in, int ln_slots, int ln_rms, float ln_eps,
                      const float *ln_colsum, const float *ln_bias, vllm_stream_t stream);

/* B5: InternRMSNorm / apex FusedRMSNorm (modeling_intern_vit.py:33-58): y = w * bf16(x * rsqrt(mean(x^2)+eps)).
 * Row strides allow the in-place QK-RMSNorm over the q / k column blocks of the qkv buffer (:131-134). */
int vllm_rmsnorm_bf16(const uint16_t *x, int ldx, const uint16_t *weight, uint16_t *y, int ldy,
                      long rows, int C, float eps, vllm_stream_t stream);
/* nn.LayerNorm (CLIP pre_layrnorm / layer_norm1,2; vl_bridge LayerNorm, modeling_visionllmv2.py:166-167). */
int vllm_layernorm_bf16(const uint16_t *x, int ldx, const uint16_t *weight, const uint16_t *bias,
                        uint16_t *y, int ldy, long rows, int C, float eps, vllm_stream_t stream);

/* B4: FlashAttention.forward(qkv[B,S,3,H,D]) -> out[B,S,H,D], non-causal, no mask, dropout 0
 * (visionllmv2/model/internvit/flash_attention.py:30-75).  D in {64,128}. */
int vllm_attn_fwd_qkvpacked_bf16(const uint16_t *qkv, uint16_t *out, int B, int S, int H, int D,
                                 float softmax_scale, vllm_stream_t stream);
/* The same for IEEE-half qkv / out (round 4): the reference's module accepts both dtypes (`assert qkv.dtype in [torch.float16,
 * torch.bfloat16]`, flash_attention.py:39-41).  fp32 scores / softmax / accumulation; P is rounded to half before P V. */
int vllm_attn_fwd_qkvpacked_f16(const uint16_t *qkv, uint16_t *out, int B, int S, int H, int D,
                                float softmax_scale, vllm_stream_t stream);

/* Patch gather for the embedding GEMM: pixels [N,3,img,img] (bf16, or fp32 when pixel_is_f32) ->
 * A [N*(img/patch)^2, Kpad] bf16, k = c*patch^2 + ky*patch + kx, zero padded to Kpad. */
int vllm_im2col_patches(const void *pixels, int pixel_is_f32, uint16_t *A, int N, int img, int patch,
                        int Kpad, vllm_stream_t stream);

/* pixel_shuffle(scale 0.5) of modeling_visionllmv2.py:381-392 applied to hidden[:, tok0:]:
 * hidden [N, tok0+hw*hw, C] (tile stride / row stride in elements, tok0 = 1 skips CLS) -> out [N, (hw/2)^2, 4C]. */
int vllm_pixel_shuffle_bf16(const uint16_t *hidden, long tile_stride, int ld, int tok0, uint16_t *out,
                            int N, int hw, int C, vllm_stream_t stream);

/* Visual-token splice (modeling_visionllmv2.py:582-605): dst[idx[i], :] = src[i, :] for i < n.  `idx` (device int64)
 * holds the flattened [B*L_txt] positions of the <im_patch> slots; rows with idx outside [0, dst_rows) are skipped. */
int vllm_scatter_rows_bf16(const uint16_t *src, const int64_t *idx, uint16_t *dst, long n, int C, long dst_rows,
                           vllm_stream_t stream);

/* The same splice with the index bookkeeping on the device (round 5): no host synchronisation, nothing copied twice.
 *   input_ids [B, L] (device int64), image_features [n_tiles, T, C] bf16 in tile order, tiles_per_sample: HOST int32 [B] ('anyres'
 *   list input; travels as a kernel argument) or NULL (one tile per sample), inputs_embeds [B, L, C] bf16, modified in place.
 * Replaces modeling_visionllmv2.py:582-605: `selected = input_ids == imp_token_id`, `has_image = selected.sum(-1) != 0` (expanded to
 * the tiles of a sample), `vit_embeds = image_features[has_image]`, `inputs_embeds[selected] = ... + vit_embeds`, and the rule of
 * :597-603 when the counts differ: slots a whole multiple of the tokens -> the tokens repeat; any other mismatch is the reference's
 * second failing assignment: NOTHING is written and status[2] = 1.
 * workspace: device int32 [vllm_splice_workspace_ints(B, L, n_tiles)]; its first four words are {rows moved, visual tokens offered
 * (tiles of samples with an image x T), error, <im_patch> slots}; status: device int32 [4] or NULL -> {slots, tokens offered, error,
 * tiles kept}. */
long vllm_splice_workspace_ints(int B, int L, int n_tiles);
int vllm_splice_visual_tokens_bf16(const int64_t *input_ids, long imp_token_id, const uint16_t *image_features,
                                   const int32_t *tiles_per_sample, int B, int L, int n_tiles, int T, int C,
                                   uint16_t *inputs_embeds, int32_t *workspace, int32_t *status, vllm_stream_t stream);

/* The per-sample token loops around the LLM (modeling_visionllmv2.py:440-527 [EMB] splice, :609-715 region features and
 * <region> slots, :775-787 [EMB] hidden states -> text_query) as index bookkeeping + ONE row mover:
 * dst[dst_idx[i], :] = src[src_idx[i], :] for i < n (device int64 indices; NULL = the identity; rows whose index falls
 * outside [0, src_rows) / [0, dst_rows) are skipped).  Rows are C bf16, C % 8 == 0. */
int vllm_copy_rows_bf16(const uint16_t *src, const int64_t *src_idx, uint16_t *dst, const int64_t *dst_idx, long n,
                        int C, long src_rows, long dst_rows, vllm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * B1. Vision encoder (the `vis_encoder` slot): one call runs patch-embed + all layers.
 *
 * Replaces InternVisionModel.forward (modeling_intern_vit.py:305-343: embeddings :82-90, encoder loop
 * :253-270, layer :198-210) and transformers.CLIPVisionModel.forward (call sites
 * visionllmv2/model/modeling_visionllmv2.py:135, 565-568).
 * ------------------------------------------------------------------------------------------------ */
#define VLLM_ARCH_INTERNVIT 0
#define VLLM_ARCH_CLIP 1

typedef struct VllmVitLayer {
    const uint16_t *norm1_w, *norm1_b;   /* RMSNorm weight (InternViT, b NULL) / LayerNorm weight+bias (CLIP) */
    const uint16_t *qkv_w, *qkv_b;       /* [3C, C] fused (CLIP: q,k,v projections concatenated), bias may be NULL */
    const uint16_t *q_norm_w, *k_norm_w; /* QK-RMSNorm weights [C] or NULL */
    const uint16_t *proj_w, *proj_b;     /* [C, C], [C] */
    const uint16_t *ls1;                 /* LayerScale [C] or NULL */
    const uint16_t *norm2_w, *norm2_b;
    const uint16_t *fc1_w, *fc1_b;       /* [I, C], [I] */
    const uint16_t *fc2_w, *fc2_b;       /* [C, I], [C] */
    const uint16_t *ls2;
    /* Optional (all NULL = launch the norms): the norms folded into the GEMMs around them (vllm_gemm_bf16_ln; taken for
     * hidden == 1024 and >= 1024 tokens).  *_w_ln = the weight with the norm's gamma multiplied in, bf16, same shape;
     * *_colsum[n] = sum_k w_ln[n, k], fp32 (LayerNorm; NULL for RMSNorm); *_bias_ln[n] = b_n + sum_k beta_k w[n, k], fp32
     * (NULL = 0).  Prepared once per weight set by the caller (the Python mirrors do it when they pack the parameters). */
    const uint16_t *qkv_w_ln;
    const float *qkv_colsum, *qkv_bias_ln;
    const uint16_t *fc1_w_ln;
    const float *fc1_colsum, *fc1_bias_ln;
} VllmVitLayer;

typedef struct VllmVitDesc {
    int arch;            /* VLLM_ARCH_* */
    int num_layers;      /* layers to run (<= model depth; hidden_states has num_layers+1 entries) */
    int hidden;          /* C */
    int heads;           /* H (C/H in {64,128}) */
    int inter;           /* I */
    int patch;           /* 14 */
    int image;           /* 336 / 448 */
    int kpad;            /* padded K of the patch-embedding GEMM (multiple of 64, >= 3*patch^2) */
    int act;             /* VLLM_EPI_GELU or VLLM_EPI_QUICK_GELU */
    int pixel_is_f32;    /* pixel_values dtype: 0 bf16, 1 fp32 */
    float eps;
    const uint16_t *patch_w;   /* [C, kpad] (Conv2d weight flattened, zero padded) */
    const uint16_t *patch_b;   /* [C] or NULL (CLIP) */
    const uint16_t *cls;       /* [C] */
    const uint16_t *pos;       /* [1+P, C] */
    const uint16_t *pre_ln_w, *pre_ln_b;  /* CLIP pre_layrnorm or NULL */
    const VllmVitLayer *layers;           /* HOST array [num_layers] of device pointers */
} VllmVitDesc;

int vllm_vit_desc_sizeof(void);   /* sizeof(VllmVitDesc): lets the ctypes mirror verify its layout */
int vllm_vit_layer_sizeof(void);
/* Workspace bytes needed by vllm_vit_forward for n_tiles tiles. */
long vllm_vit_workspace_bytes(const VllmVitDesc *desc, int n_tiles);
/* pixels [n_tiles,3,image,image]; hidden_states: HOST array of num_layers+1 DEVICE pointers, each
 * [n_tiles, 1+P, C] bf16 contiguous (entry i = input of layer i, last = final output, exactly the tuple
 * InternVisionEncoder returns with output_hidden_states=True).  Entries may be NULL except the last: the
 * library then keeps that state in its workspace (nobody reads it). */
int vllm_vit_forward(const VllmVitDesc *desc, const void *pixels, int n_tiles,
                     uint16_t *const *hidden_states, void *workspace, long workspace_bytes,
                     vllm_stream_t stream);
/* Cumulative number of GEMM launches of vllm_vit_forward that ran with a norm folded in (test hook: proves which path ran). */
long vllm_vit_folded_gemm_launches(void);

/* ------------------------------------------------------------------------------------------------
 * B2. Projector ("vl_bridge") with the hidden-state select / CLS drop / pixel-shuffle in front of it
 * (modeling_visionllmv2.py:569-579, 162-182, 381-392).
 * ------------------------------------------------------------------------------------------------ */
#define VLLM_BRIDGE_LINEAR 0        /* nn.Linear */
#define VLLM_BRIDGE_MLP_GELU 1      /* mlp{N}x_gelu: Linear (GELU Linear)*(depth-1) */
#define VLLM_BRIDGE_INTERNVL_MLP 2  /* LayerNorm, Linear, GELU, Linear */

typedef struct VllmBridgeDesc {
    int kind;             /* VLLM_BRIDGE_* */
    int depth;            /* number of Linear layers (1 for linear, N for mlpNx_gelu, 2 for internvl_mlp) */
    int in_features;      /* C or 4C */
    int out_features;     /* LLM hidden size */
    int pixel_shuffle;    /* 1: apply pixel_shuffle(0.5) to the token grid first */
    int skip_cls;         /* 1: `hidden` is [n, 1+T, C] and row 0 of every tile (CLS) is skipped (:571); 0: [n, T, C] */
    float ln_eps;
    const uint16_t *ln_w, *ln_b;     /* internvl_mlp LayerNorm or NULL */
    const uint16_t *w[4], *b[4];     /* Linear weights [out,in] / biases */
} VllmBridgeDesc;

int vllm_bridge_desc_sizeof(void);
long vllm_bridge_workspace_bytes(const VllmBridgeDesc *desc, int n_tiles, int tokens_per_tile_in);
/* hidden: the selected hidden state [n_tiles, skip_cls+T, C] bf16; out
 * [n_tiles, T or T/4, out_features] bf16 -- row-major [tile, token, C_llm], the layout the token splice
 * (modeling_visionllmv2.py:582-605) consumes. */
int vllm_bridge_forward(const VllmBridgeDesc *desc, const uint16_t *hidden, int n_tiles, int T, int C,
                        uint16_t *out, void *workspace, long workspace_bytes, vllm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLLM_HIP_H */
