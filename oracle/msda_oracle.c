/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the reference's multi-scale deformable attention
 * (MSDA) operator.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * Follows (reference paths relative to /root/reference/VisionLLMv2/):
 *   forward  : visionllmv2/model/unipose/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-298
 *              (index decode, h_im = loc_h*H - 0.5, acceptance test
 *               h_im > -1 && w_im > -1 && h_im < H && w_im < W)
 *   bilinear : same file :33-84 (floor, corner guards, weights hh*hw, hh*lw, lh*hw, lh*lw)
 *   backward : same file :87-161 (col2im bilinear: grad_value scatter, grad_sampling_loc,
 *              grad_attn_weight) and :301-360 (per-(b,q,m,c) loop, reduction over c)
 *   mmcv twin: mmcv/mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:201-255
 *
 * Parity pinning: checked in tests/test_oracle_msda.py against
 *   - the reference's own known-answer inputs (mmcv/tests/test_ops/test_ms_deformable_attn.py:53-134,
 *     torch.manual_seed(3)), with outputs produced by the reference's
 *     multi_scale_deformable_attn_pytorch imported in the build container
 *     (fixtures: tests/golden/msda_*.npz, generator: oracle/gen_golden.py).
 *
 * Arithmetic notes.  The reference kernel is templated on scalar_t (float/double).
 * `loc_h * spatial_h - 0.5` is evaluated as round(round(loc_h*H) - 0.5) in scalar_t
 * (the multiply cannot be fused with the subtraction: the literal 0.5 is a double, so
 * the product is materialised first).  We therefore compile this file with
 * -ffp-contract=off and write the two roundings explicitly.  floor()/comparisons are
 * the integer ("index-exact") part of the contract.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define DEFINE_MSDA(SUFFIX, T, FLOORF)                                                          \
                                                                                                \
/* One sampling point: integer part.  Returns 1 if the point is accepted. */                   \
static inline int msda_point_##SUFFIX(T loc_w, T loc_h, int H, int W,                           \
                                      T *h_im_o, T *w_im_o, int *h_low_o, int *w_low_o)         \
{                                                                                               \
    volatile T ph = loc_h * (T)H; /* volatile: forbid contraction whatever the flags */        \
    volatile T pw = loc_w * (T)W;                                                               \
    const T h_im = ph - (T)0.5;                                                                 \
    const T w_im = pw - (T)0.5;                                                                 \
    *h_im_o = h_im; *w_im_o = w_im;                                                             \
    *h_low_o = (int)FLOORF(h_im);                                                               \
    *w_low_o = (int)FLOORF(w_im);                                                               \
    return (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) ? 1 : 0;               \
}                                                                                               \
                                                                                                \
void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *lsi,   \
                                  const T *loc, const T *attw,                                  \
                                  int B, int S, int M, int D, int L, int Lq, int P, T *out)     \
{                                                                                               \
    const long qid_stride = (long)M * D;                                                        \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                    \
    for (int b = 0; b < B; ++b)                                                                 \
    for (int q = 0; q < Lq; ++q) {                                                              \
        for (int m = 0; m < M; ++m) {                                                           \
            const long samp = ((long)b * Lq + q) * M + m;                                       \
            T *o = out + samp * D;                                                              \
            for (int c = 0; c < D; ++c) o[c] = (T)0;                                            \
            long wptr = samp * L * P;                                                           \
            long lptr = wptr << 1;                                                              \
            for (int l = 0; l < L; ++l) {                                                       \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                   \
                const T *vbase = value + ((long)b * S + (long)lsi[l]) * qid_stride;             \
                for (int p = 0; p < P; ++p, ++wptr, lptr += 2) {                                \
                    T h_im, w_im; int h_low, w_low;                                             \
                    if (!msda_point_##SUFFIX(loc[lptr], loc[lptr + 1], H, W,                    \
                                             &h_im, &w_im, &h_low, &w_low)) continue;           \
                    const int h_high = h_low + 1, w_high = w_low + 1;                           \
                    const T lh = h_im - (T)h_low, lw = w_im - (T)w_low;                         \
                    const T hh = (T)1 - lh, hw = (T)1 - lw;                                     \
                    const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;             \
                    const T aw = attw[wptr];                                                    \
                    const long w_stride = qid_stride, h_stride = (long)W * w_stride;            \
                    const T *p1 = (h_low >= 0 && w_low >= 0)                                    \
                        ? vbase + h_low * h_stride + w_low * w_stride + (long)m * D : NULL;     \
                    const T *p2 = (h_low >= 0 && w_high <= W - 1)                               \
                        ? vbase + h_low * h_stride + w_high * w_stride + (long)m * D : NULL;    \
                    const T *p3 = (h_high <= H - 1 && w_low >= 0)                               \
                        ? vbase + h_high * h_stride + w_low * w_stride + (long)m * D : NULL;    \
                    const T *p4 = (h_high <= H - 1 && w_high <= W - 1)                          \
                        ? vbase + h_high * h_stride + w_high * w_stride + (long)m * D : NULL;   \
                    for (int c = 0; c < D; ++c) {                                               \
                        const T v1 = p1 ? p1[c] : (T)0, v2 = p2 ? p2[c] : (T)0;                 \
                        const T v3 = p3 ? p3[c] : (T)0, v4 = p4 ? p4[c] : (T)0;                 \
                        const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                  \
                        o[c] += val * aw;                                                       \
                    }                                                                           \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
    }                                                                                           \
}                                                                                               \
                                                                                                \
/* Integer part of the sampling, for index-exact parity: per point (b,q,m,l,p):               \
 * h_low, w_low (int32) and a mask byte: bit0 accepted, bit1..4 corners 1..4 in bounds. */     \
void msda_oracle_sample_index_##SUFFIX(const int64_t *shapes, const T *loc,                    \
                                       int B, int M, int L, int Lq, int P,                      \
                                       int32_t *h_low_o, int32_t *w_low_o, uint8_t *mask_o)     \
{                                                                                               \
    const long n = (long)B * Lq * M;                                                            \
    for (long s = 0; s < n; ++s)                                                                \
        for (int l = 0; l < L; ++l) {                                                           \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                       \
            for (int p = 0; p < P; ++p) {                                                       \
                const long i = (s * L + l) * P + p;                                             \
                T h_im, w_im; int h_low, w_low;                                                 \
                const int ok = msda_point_##SUFFIX(loc[2 * i], loc[2 * i + 1], H, W,            \
                                                   &h_im, &w_im, &h_low, &w_low);               \
                uint8_t mk = (uint8_t)ok;                                                       \
                if (ok) {                                                                       \
                    const int h_high = h_low + 1, w_high = w_low + 1;                           \
                    if (h_low >= 0 && w_low >= 0) mk |= 2;                                      \
                    if (h_low >= 0 && w_high <= W - 1) mk |= 4;                                 \
                    if (h_high <= H - 1 && w_low >= 0) mk |= 8;                                 \
                    if (h_high <= H - 1 && w_high <= W - 1) mk |= 16;                           \
                } else { h_low = 0; w_low = 0; }                                                \
                h_low_o[i] = h_low; w_low_o[i] = w_low; mask_o[i] = mk;                         \
            }                                                                                   \
        }                                                                                       \
}                                                                                               \
                                                                                                \
/* Backward.  grad_value / grad_loc / grad_attw must be zero-filled by the caller             \
 * (the reference allocates them with at::zeros, ms_deform_attn_cuda.cu:118-120).             \
 * Deterministic sequential accumulation (the reference uses atomics). */                      \
void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *lsi,  \
                                   const T *loc, const T *attw, const T *grad_out,              \
                                   int B, int S, int M, int D, int L, int Lq, int P,            \
                                   T *grad_value, T *grad_loc, T *grad_attw)                    \
{                                                                                               \
    const long qid_stride = (long)M * D;                                                        \
    for (int b = 0; b < B; ++b)                                                                 \
    for (int q = 0; q < Lq; ++q)                                                                \
    for (int m = 0; m < M; ++m) {                                                               \
        const long samp = ((long)b * Lq + q) * M + m;                                           \
        const T *go = grad_out + samp * D;                                                      \
        long wptr = samp * L * P;                                                               \
        long lptr = wptr << 1;                                                                  \
        for (int l = 0; l < L; ++l) {                                                           \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                       \
            const long lvl_off = ((long)b * S + (long)lsi[l]) * qid_stride;                     \
            const T *vbase = value + lvl_off;                                                   \
            T *gvbase = grad_value + lvl_off;                                                   \
            for (int p = 0; p < P; ++p, ++wptr, lptr += 2) {                                    \
                T h_im, w_im; int h_low, w_low;                                                 \
                if (!msda_point_##SUFFIX(loc[lptr], loc[lptr + 1], H, W,                        \
                                         &h_im, &w_im, &h_low, &w_low)) continue;               \
                const int h_high = h_low + 1, w_high = w_low + 1;                               \
                const T lh = h_im - (T)h_low, lw = w_im - (T)w_low;                             \
                const T hh = (T)1 - lh, hw = (T)1 - lw;                                         \
                const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                 \
                const T aw = attw[wptr];                                                        \
                const long w_stride = qid_stride, h_stride = (long)W * w_stride;                \
                const long o1 = h_low * h_stride + w_low * w_stride + (long)m * D;              \
                const long o2 = h_low * h_stride + w_high * w_stride + (long)m * D;             \
                const long o3 = h_high * h_stride + w_low * w_stride + (long)m * D;             \
                const long o4 = h_high * h_stride + w_high * w_stride + (long)m * D;            \
                const int k1 = (h_low >= 0 && w_low >= 0), k2 = (h_low >= 0 && w_high <= W - 1);\
                const int k3 = (h_high <= H - 1 && w_low >= 0);                                 \
                const int k4 = (h_high <= H - 1 && w_high <= W - 1);                            \
                T g_aw = (T)0, g_x = (T)0, g_y = (T)0;                                          \
                for (int c = 0; c < D; ++c) {                                                   \
                    const T top_grad = go[c];                                                   \
                    const T tgv = top_grad * aw;                                                \
                    T ghw = (T)0, gww = (T)0;                                                   \
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                           \
                    if (k1) { v1 = vbase[o1 + c]; ghw -= hw * v1; gww -= hh * v1;               \
                              gvbase[o1 + c] += w1 * tgv; }                                     \
                    if (k2) { v2 = vbase[o2 + c]; ghw -= lw * v2; gww += hh * v2;               \
                              gvbase[o2 + c] += w2 * tgv; }                                     \
                    if (k3) { v3 = vbase[o3 + c]; ghw += hw * v3; gww -= lh * v3;               \
                              gvbase[o3 + c] += w3 * tgv; }                                     \
                    if (k4) { v4 = vbase[o4 + c]; ghw += lw * v4; gww += lh * v4;               \
                              gvbase[o4 + c] += w4 * tgv; }                                     \
                    const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                      \
                    g_aw += top_grad * val;                                                     \
                    g_x += (T)W * gww * tgv;                                                    \
                    g_y += (T)H * ghw * tgv;                                                    \
                }                                                                               \
                grad_attw[wptr] += g_aw;                                                        \
                grad_loc[lptr] += g_x;                                                          \
                grad_loc[lptr + 1] += g_y;                                                      \
            }                                                                                   \
        }                                                                                       \
    }                                                                                           \
}

DEFINE_MSDA(f32, float, floorf)
DEFINE_MSDA(f64, double, floor)

int msda_oracle_abi_version(void) { return 1; }
