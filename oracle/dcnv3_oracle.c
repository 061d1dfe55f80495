/* ORACLE -- test infrastructure only.  Plain-C restatement of the DCNv3 forward of the reference
 * (VisionLLMv2/visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh):
 *   dcnv3_im2col_bilinear      :31-84    bilinear sample with zero padding of a [H, W, G, C] image
 *   dcnv3_im2col_gpu_kernel    :217-278  index decode c -> g -> w_out -> h_out -> b, reference point p0, the
 *                                        kernel_w-outer / kernel_h-inner point order, the acceptance test
 *   output geometry            dcnv3_cuda.cu:40-45
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

DEFINE_DCNV3(float, f32, floorf)
DEFINE_DCNV3(double, f64, floor)
