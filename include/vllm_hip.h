/*
 * vllm_hip.h -- C ABI of libvllm_hip.so: the MI355X (gfx950) image->visual-token hot path of VisionLLMv2.
 *
 * Plain pointers and sizes only (no torch / ATen types).  Every pointer marked "device" is a HIP device
 * pointer owned by the caller (PyTorch's caching allocator in practice); the library never allocates or
 * frees device memory and never synchronises: all work is enqueued on `stream` (a hipStream_t passed as
 * void*; NULL = the legacy default stream).  Thread-safety: entry points are re-entrant; the only global
 * state is the per-thread last-error string.
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

#define VLLM_ABI_VERSION 1

#define VLLM_OK 0
#define VLLM_EINVAL (-1)   /* bad argument (shape / alignment / unsupported size) */
#define VLLM_ELAUNCH (-2)  /* hipLaunchKernel / HIP runtime error */
#define VLLM_ENOTIMPL (-3)

typedef void *vllm_stream_t; /* hipStream_t */

int vllm_abi_version(void);
const char *vllm_last_error(void);
/* Fills name[0..cap) with the device's gcnArchName; returns CU count or negative error. */
int vllm_device_info(char *name, int cap);

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
int vllm_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                           const double *loc, const double *attw, const double *grad_out,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           double *grad_value, double *grad_loc, double *grad_attw, vllm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLLM_HIP_H */
