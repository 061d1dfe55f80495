/* ORACLE -- test infrastructure only.  Plain-C restatement of the DCNv3 forward of the reference
 * (VisionLLMv2/visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh):
 *   dcnv3_im2col_bilinear      :31-84    bilinear sample with zero padding of a [H, W, G, C] image
 *   dcnv3_im2col_gpu_kernel    :217-278  index decode c -> g -> w_out -> h_out -> b, reference point p0, the
 *                                        kernel_w-outer / kernel_h-inner point order, the acceptance test
 *   output geometry            dcnv3_cuda.cu:40-45
 *   dcnv3_col2im_bilinear      :86-146   backward of one sampling point: atomicAdd of w_i * top_grad * mask into grad_im at the
 *                                        four corners (per-corner bounds), grad_mask = top_grad * val, grad_offset = offset_scale *
 *                                        grad_{w,h}_weight * top_grad * mask (x first, then y)
 *   dcnv3_col2im_gpu_kernel_*  :279-857  all variants: the per-channel contributions of a (pixel, group, point) are SUMMED over the
 *                                        group's channels (shared-memory reductions :351-366) into grad_offset / grad_mask
 * Compiled by oracle/Makefile (gcc, -ffp-contract=off).  Pinned by tests/golden/dcnv3_*.npz, which
 * oracle/gen_golden.py produces by RUNNING the reference's dcnv3_core_pytorch (functions/dcnv3_func.py:121-161) on the
 * inputs of the reference's own test (ops_dcnv3/test.py:19-66, seed 3). */
#include <math.h>
#include <stdint.h>

#define DEFINE_DCNV3(T, SUFFIX, FLOOR)                                                                              \
    static T bilinear_##SUFFIX(const T *im, int H, int W, int G, int C, T h, T w, int g, int c)                     \
    {                                                                                                               \
        const int h_low = (int)FLOOR(h), w_low = (int)FLOOR(w);                                                     \
        const int h_high = h_low + 1, w_high = w_low + 1;                                                           \
        const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                           \
        const long ws = (long)G * C, hs = (long)W * ws, base = (long)g * C + c;                                     \
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                           \
        if (h_low >= 0 && w_low >= 0) v1 = im[h_low * hs + w_low * ws + base];                                      \
        if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * hs + w_high * ws + base];                                \
        if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * hs + w_low * ws + base];                                \
        if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * hs + w_high * ws + base];                          \
        const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                             \
        return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                                               \
    }                                                                                                               \
    int dcnv3_forward_##SUFFIX(const T *input, const T *offset, const T *mask, int N, int H, int W, int G, int C,   \
                               int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, T offset_scale,     \
                               int Ho, int Wo, T *out)                                                             \
    {                                                                                                               \
        const int K = kh * kw;                                                                                      \
        _Pragma("omp parallel for collapse(2)") for (int b = 0; b < N; ++b) for (int y = 0; y < Ho; ++y)            \
            for (int x = 0; x < Wo; ++x)                                                                            \
                for (int g = 0; g < G; ++g) {                                                                       \
                    const long sidx = (((long)b * Ho + y) * Wo + x) * G + g;                                        \
                    const int p0_w = ((dw * (kw - 1)) >> 1) - pw + x * sw;                                          \
                    const int p0_h = ((dh * (kh - 1)) >> 1) - ph + y * sh;                                          \
                    const T p0_w_ = p0_w - ((dw * (kw - 1)) >> 1) * offset_scale;                                   \
                    const T p0_h_ = p0_h - ((dh * (kh - 1)) >> 1) * offset_scale;                                   \
                    const T *im = input + (long)b * H * W * G * C;                                                  \
                    for (int c = 0; c < C; ++c) {                                                                   \
                        T col = 0;                                                                                  \
                        long wp = sidx * K, lp = wp * 2;                                                            \
                        for (int i = 0; i < kw; ++i)                                                                \
                            for (int j = 0; j < kh; ++j) {                                                          \
                                const T off_w = offset[lp], off_h = offset[lp + 1];                                 \
                                const T loc_w = p0_w_ + (i * dw + off_w) * offset_scale;                            \
                                const T loc_h = p0_h_ + (j * dh + off_h) * offset_scale;                            \
                                const T wgt = mask[wp];                                                             \
                                if (loc_h > -1 && loc_w > -1 && loc_h < H && loc_w < W)                             \
                                    col += bilinear_##SUFFIX(im, H, W, G, C, loc_h, loc_w, g, c) * wgt;             \
                                wp += 1;                                                                            \
                                lp += 2;                                                                            \
                            }                                                                                       \
                        out[sidx * C + c] = col;                                                                    \
                    }                                                                                               \
                }                                                                                                   \
        return 0;                                                                                                   \
    }

/* backward (round 4).  grad_input must be zero-filled by the caller (the reference: at::zeros_like, dcnv3_cuda.cu:118);      \
 * grad_offset / grad_mask are written completely.  Serial over (b, y, x) so that the sums into grad_input have a fixed order. */
#define DEFINE_DCNV3_BWD(T, SUFFIX, FLOOR)                                                                          \
    int dcnv3_backward_##SUFFIX(const T *input, const T *offset, const T *mask, const T *grad_out, int N, int H, int W,   \
                                int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,       \
                                T offset_scale, int Ho, int Wo, T *grad_input, T *grad_offset, T *grad_mask)        \
    {                                                                                                               \
        const int K = kh * kw;                                                                                      \
        const long ws = (long)G * C, hs = (long)W * ws;                                                             \
        for (int b = 0; b < N; ++b) for (int y = 0; y < Ho; ++y) for (int x = 0; x < Wo; ++x)                       \
            for (int g = 0; g < G; ++g) {                                                                           \
                const long sidx = (((long)b * Ho + y) * Wo + x) * G + g;                                            \
                const int p0_w = ((dw * (kw - 1)) >> 1) - pw + x * sw;                                              \
                const int p0_h = ((dh * (kh - 1)) >> 1) - ph + y * sh;                                              \
                const T p0_w_ = p0_w - ((dw * (kw - 1)) >> 1) * offset_scale;                                       \
                const T p0_h_ = p0_h - ((dh * (kh - 1)) >> 1) * offset_scale;                                       \
                const T *im = input + (long)b * H * hs;                                                             \
                T *gim = grad_input + (long)b * H * hs;                                                             \
                long wp = sidx * K, lp = wp * 2;                                                                    \
                for (int i = 0; i < kw; ++i)                                                                        \
                    for (int j = 0; j < kh; ++j) {                                                                  \
                        const T off_w = offset[lp], off_h = offset[lp + 1];                                         \
                        const T loc_w = p0_w_ + (i * dw + off_w) * offset_scale;                                    \
                        const T loc_h = p0_h_ + (j * dh + off_h) * offset_scale;                                    \
                        const T wgt = mask[wp];                                                                     \
                        T g_w = 0, g_h = 0, g_a = 0;                                                                \
                        if (loc_h > -1 && loc_w > -1 && loc_h < H && loc_w < W) {                                   \
                            const int h_low = (int)FLOOR(loc_h), w_low = (int)FLOOR(loc_w);                         \
                            const int h_high = h_low + 1, w_high = w_low + 1;                                       \
                            const T lh = loc_h - h_low, lw = loc_w - w_low, hh = 1 - lh, hw = 1 - lw;               \
                            const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                         \
                            for (int c = 0; c < C; ++c) {                                                           \
                                const long base = (long)g * C + c;                                                  \
                                const T top = grad_out[sidx * C + c], top_im = top * wgt;                           \
                                T ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                 \
                                if (h_low >= 0 && w_low >= 0) {                                                     \
                                    const long p1 = h_low * hs + w_low * ws + base;                                 \
                                    v1 = im[p1]; ghw -= hw * v1; gww -= hh * v1; gim[p1] += w1 * top_im;            \
                                }                                                                                   \
                                if (h_low >= 0 && w_high <= W - 1) {                                                \
                                    const long p2 = h_low * hs + w_high * ws + base;                                \
                                    v2 = im[p2]; ghw -= lw * v2; gww += hh * v2; gim[p2] += w2 * top_im;            \
                                }                                                                                   \
                                if (h_high <= H - 1 && w_low >= 0) {                                                \
                                    const long p3 = h_high * hs + w_low * ws + base;                                \
                                    v3 = im[p3]; ghw += hw * v3; gww -= lh * v3; gim[p3] += w3 * top_im;            \
                                }                                                                                   \
                                if (h_high <= H - 1 && w_high <= W - 1) {                                           \
                                    const long p4 = h_high * hs + w_high * ws + base;                               \
                                    v4 = im[p4]; ghw += lw * v4; gww += lh * v4; gim[p4] += w4 * top_im;            \
                                }                                                                                   \
                                const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                \
                                g_a += top * val;                                                                   \
                                g_w += offset_scale * gww * top_im;                                                 \
                                g_h += offset_scale * ghw * top_im;                                                 \
                            }                                                                                       \
                        }                                                                                           \
                        grad_mask[wp] = g_a;                                                                        \
                        grad_offset[lp] = g_w;                                                                      \
                        grad_offset[lp + 1] = g_h;                                                                  \
                        wp += 1;                                                                                    \
                        lp += 2;                                                                                    \
                    }                                                                                               \
            }                                                                                                       \
        return 0;                                                                                                   \
    }

DEFINE_DCNV3(float, f32, floorf)
DEFINE_DCNV3(double, f64, floor)
DEFINE_DCNV3_BWD(float, f32, floorf)
DEFINE_DCNV3_BWD(double, f64, floor)
