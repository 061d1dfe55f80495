// DCNv3 backward (round 4: the row after SURVEY section 8-f3).
// Reference: visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:86-146 (dcnv3_col2im_bilinear: the backward of ONE sampling
// point -- atomicAdd of w_i * top_grad * mask into grad_im at the four corners under per-corner bounds, grad_mask = top_grad * val,
// grad_offset = offset_scale * grad_{w,h}_weight * top_grad * mask, x first), :279-857 (six col2im kernel variants that differ in
// how the per-channel contributions of a (pixel, group, point) are reduced over the group's channels), dcnv3_cuda.cu:92-174 (host:
// grad_input zero-filled, one launch per im2col_step slice), functions/dcnv3_func.py:51-59 (DCNv3Function.backward).
//
// Two kernels:
//   * dcnv3_bwd_vec_kernel<T, LPG>: group channels = 4 * LPG with LPG in {1, 2, 4, 8, 16}.  The forward gather kernel's mapping: a
//     lane owns 4 consecutive channels of one (batch, output pixel, group), the LPG lanes of a (pixel, group) sit next to each
//     other in a wave.  Per point a lane loads its 16 bytes of the four corners (clamped addresses, selects decide what counts),
//     scatters w_i * top_grad * mask with hardware floating-point atomics (4 per corner), and the three per-point sums over the
//     group's channels are reduced with DPP-free cross-lane shuffles inside the LPG-lane group (a fixed tree: deterministic);
//     lane 0 of the group stores grad_offset / grad_mask.  The scatter order (and with it the last bits of grad_input) is the
//     hardware's, as in the reference.
//   * dcnv3_bwd_generic_kernel<T>: any channel count (the reference's gradcheck list has 1, 30, 71, 1025): one thread per
//     (batch, output pixel, group) walks the channels -- grad_offset / grad_mask are plain sequential sums, grad_input atomics.
// grad_input must be zero-filled by the caller (the Python mirror allocates it with torch.zeros, as the reference's host code
// does with at::zeros_like); grad_offset / grad_mask are written completely (rejected points: zeros).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "dcnv3_geo.hpp"

namespace vllm {

namespace {

template <typename T> __device__ __forceinline__ T bwd_floor(T x);
template <> __device__ __forceinline__ float bwd_floor<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double bwd_floor<double>(double x) { return floor(x); }

// the geometry of one sampling point, shared by both kernels: acceptance (dcnv3_im2col_cuda.cuh:335-336), integer part that
// never sees a rejected (possibly NaN / inf) coordinate, bilinear fractions, per-corner validity, clamped corner rows / columns
template <typename T>
struct BwdPoint {
    bool ok, k1, k2, k3, k4;
    int y0, y1, x0, x1;
    T lh, lw, hh, hw;
};
template <typename T>
__device__ __forceinline__ BwdPoint<T> bwd_point(T loc_h, T loc_w, int H, int W)
{
    BwdPoint<T> p;
    p.ok = loc_h > (T)-1 && loc_w > (T)-1 && loc_h < (T)H && loc_w < (T)W;
    const int h_low = p.ok ? (int)bwd_floor<T>(loc_h) : 0, w_low = p.ok ? (int)bwd_floor<T>(loc_w) : 0;
    p.lh = p.ok ? loc_h - (T)h_low : (T)0;
    p.lw = p.ok ? loc_w - (T)w_low : (T)0;
    p.hh = (T)1 - p.lh;
    p.hw = (T)1 - p.lw;
    const bool u0 = h_low >= 0, u1 = h_low + 1 <= H - 1, l0 = w_low >= 0, l1 = w_low + 1 <= W - 1;
    p.k1 = p.ok && u0 && l0; p.k2 = p.ok && u0 && l1; p.k3 = p.ok && u1 && l0; p.k4 = p.ok && u1 && l1;
    p.y0 = min(max(h_low, 0), H - 1); p.y1 = min(max(h_low + 1, 0), H - 1);
    p.x0 = min(max(w_low, 0), W - 1); p.x1 = min(max(w_low + 1, 0), W - 1);
    return p;
}

template <typename T, int LPG>
__global__ __launch_bounds__(256) void dcnv3_bwd_vec_kernel(const T *__restrict__ in, const T *__restrict__ off,
                                                            const T *__restrict__ msk, const T *__restrict__ gout,
                                                            T *__restrict__ gin, T *__restrict__ goff, T *__restrict__ gmsk,
                                                            long total, Dcnv3Geo q, T offset_scale)
{
    const long idx0 = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx0 < total;
    const long idx = live ? idx0 : total - 1;        // (whole LPG-lane groups take part in the shuffles: `total` is a multiple of LPG)
    const int cc = (int)(idx % LPG);
    const long sidx = idx / LPG;                      // ((b * Ho + y) * Wo + x) * G + g
    long t = sidx;
    const int g = (int)(t % q.G); t /= q.G;
    const int x = (int)(t % q.Wo); t /= q.Wo;
    const int y = (int)(t % q.Ho);
    const long b = t / q.Ho;
    const int K = q.kh * q.kw;
    const int p0_w = ((q.dw * (q.kw - 1)) >> 1) - q.pw + x * q.sw;
    const int p0_h = ((q.dh * (q.kh - 1)) >> 1) - q.ph + y * q.sh;
    const T p0_w_ = (T)p0_w - dcn_mul_rn<T>((T)((q.dw * (q.kw - 1)) >> 1), offset_scale);
    const T p0_h_ = (T)p0_h - dcn_mul_rn<T>((T)((q.dh * (q.kh - 1)) >> 1), offset_scale);
    const long ws = (long)q.G * q.C, hs = (long)q.W * ws;
    const long imo = b * q.H * hs + (long)g * q.C + cc * 4;
    const T *im = in + imo;
    T *gim = gin + imo;
    const T *op = off + sidx * K * 2;
    const T *mp = msk + sidx * K;
    T top[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) top[v] = gout[sidx * q.C + cc * 4 + v];
    long wp = sidx * K;
    for (int i = 0; i < q.kw; ++i)
        for (int j = 0; j < q.kh; ++j) {
            const T off_w = op[0], off_h = op[1], wgt = mp[0];
            op += 2; mp += 1;
            const T loc_w = dcn_loc<T>(p0_w_, (T)(i * q.dw), off_w, offset_scale);
            const T loc_h = dcn_loc<T>(p0_h_, (T)(j * q.dh), off_h, offset_scale);
            const BwdPoint<T> p = bwd_point<T>(loc_h, loc_w, q.H, q.W);
            const long o1 = p.y0 * hs + p.x0 * ws, o2 = p.y0 * hs + p.x1 * ws, o3 = p.y1 * hs + p.x0 * ws, o4 = p.y1 * hs + p.x1 * ws;
            const T w1 = p.hh * p.hw, w2 = p.hh * p.lw, w3 = p.lh * p.hw, w4 = p.lh * p.lw;
            T g_a = 0, g_w = 0, g_h = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const T v1 = p.k1 ? im[o1 + v] : (T)0, v2 = p.k2 ? im[o2 + v] : (T)0, v3 = p.k3 ? im[o3 + v] : (T)0,
                        v4 = p.k4 ? im[o4 + v] : (T)0;
                const T tg = top[v], tgi = tg * wgt;
                // (the reference's order of the four signed terms, :111-137)
                T ghw = 0, gww = 0;
                ghw -= p.hw * v1; gww -= p.hh * v1;
                ghw -= p.lw * v2; gww += p.hh * v2;
                ghw += p.hw * v3; gww -= p.lh * v3;
                ghw += p.lw * v4; gww += p.lh * v4;
                if (live) {
                    if (p.k1) unsafeAtomicAdd(gim + o1 + v, w1 * tgi);
                    if (p.k2) unsafeAtomicAdd(gim + o2 + v, w2 * tgi);
                    if (p.k3) unsafeAtomicAdd(gim + o3 + v, w3 * tgi);
                    if (p.k4) unsafeAtomicAdd(gim + o4 + v, w4 * tgi);
                }
                const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                g_a += tg * val;
                g_w += offset_scale * gww * tgi;
                g_h += offset_scale * ghw * tgi;
            }
            // sums over the group's channels: a fixed butterfly over the LPG lanes of the (pixel, group)
#pragma unroll
            for (int d = 1; d < LPG; d <<= 1) {
                g_a += __shfl_xor(g_a, d);
                g_w += __shfl_xor(g_w, d);
                g_h += __shfl_xor(g_h, d);
            }
            if (live && cc == 0) {
                gmsk[wp] = p.ok ? g_a : (T)0;
                goff[wp * 2] = p.ok ? g_w : (T)0;
                goff[wp * 2 + 1] = p.ok ? g_h : (T)0;
            }
            wp += 1;
        }
}

template <typename T>
__global__ __launch_bounds__(256) void dcnv3_bwd_generic_kernel(const T *__restrict__ in, const T *__restrict__ off,
                                                                const T *__restrict__ msk, const T *__restrict__ gout,
                                                                T *__restrict__ gin, T *__restrict__ goff, T *__restrict__ gmsk,
                                                                long total, Dcnv3Geo q, T offset_scale)
{
    const long sidx = (long)blockIdx.x * 256 + threadIdx.x;   // ((b * Ho + y) * Wo + x) * G + g
    if (sidx >= total) return;
    long t = sidx;
    const int g = (int)(t % q.G); t /= q.G;
    const int x = (int)(t % q.Wo); t /= q.Wo;
    const int y = (int)(t % q.Ho);
    const long b = t / q.Ho;
    const int K = q.kh * q.kw;
    const int p0_w = ((q.dw * (q.kw - 1)) >> 1) - q.pw + x * q.sw;
    const int p0_h = ((q.dh * (q.kh - 1)) >> 1) - q.ph + y * q.sh;
    const T p0_w_ = (T)p0_w - dcn_mul_rn<T>((T)((q.dw * (q.kw - 1)) >> 1), offset_scale);
    const T p0_h_ = (T)p0_h - dcn_mul_rn<T>((T)((q.dh * (q.kh - 1)) >> 1), offset_scale);
    const long ws = (long)q.G * q.C, hs = (long)q.W * ws;
    const long imo = b * q.H * hs + (long)g * q.C;
    const T *im = in + imo;
    T *gim = gin + imo;
    const T *tp = gout + sidx * q.C;
    long wp = sidx * K;
    for (int i = 0; i < q.kw; ++i)
        for (int j = 0; j < q.kh; ++j) {
            const T off_w = off[wp * 2], off_h = off[wp * 2 + 1], wgt = msk[wp];
            const T loc_w = dcn_loc<T>(p0_w_, (T)(i * q.dw), off_w, offset_scale);
            const T loc_h = dcn_loc<T>(p0_h_, (T)(j * q.dh), off_h, offset_scale);
            const BwdPoint<T> p = bwd_point<T>(loc_h, loc_w, q.H, q.W);
            const long o1 = p.y0 * hs + p.x0 * ws, o2 = p.y0 * hs + p.x1 * ws, o3 = p.y1 * hs + p.x0 * ws, o4 = p.y1 * hs + p.x1 * ws;
            const T w1 = p.hh * p.hw, w2 = p.hh * p.lw, w3 = p.lh * p.hw, w4 = p.lh * p.lw;
            T g_a = 0, g_w = 0, g_h = 0;
            if (p.ok)
                for (int c = 0; c < q.C; ++c) {
                    const T v1 = p.k1 ? im[o1 + c] : (T)0, v2 = p.k2 ? im[o2 + c] : (T)0, v3 = p.k3 ? im[o3 + c] : (T)0,
                            v4 = p.k4 ? im[o4 + c] : (T)0;
                    const T tg = tp[c], tgi = tg * wgt;
                    T ghw = 0, gww = 0;
                    ghw -= p.hw * v1; gww -= p.hh * v1;
                    ghw -= p.lw * v2; gww += p.hh * v2;
                    ghw += p.hw * v3; gww -= p.lh * v3;
                    ghw += p.lw * v4; gww += p.lh * v4;
                    if (p.k1) unsafeAtomicAdd(gim + o1 + c, w1 * tgi);
                    if (p.k2) unsafeAtomicAdd(gim + o2 + c, w2 * tgi);
                    if (p.k3) unsafeAtomicAdd(gim + o3 + c, w3 * tgi);
                    if (p.k4) unsafeAtomicAdd(gim + o4 + c, w4 * tgi);
                    const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                    g_a += tg * val;
                    g_w += offset_scale * gww * tgi;
                    g_h += offset_scale * ghw * tgi;
                }
            gmsk[wp] = g_a;
            goff[wp * 2] = g_w;
            goff[wp * 2 + 1] = g_h;
            wp += 1;
        }
}

template <typename T>
int dcnv3_bwd_launch(const T *in, const T *off, const T *msk, const T *gout, Dcnv3Geo q, T offset_scale, T *gin, T *goff, T *gmsk,
                     hipStream_t st)
{
    const long pix = (long)q.N * q.Ho * q.Wo * q.G;
    if (pix == 0 || q.C == 0) return VLLM_OK;
    VLLM_REQUIRE(pix * q.C < (1L << 40) && (long)q.N * q.H * q.W * q.G * q.C < (1L << 40), "dcnv3_backward: too many elements");
    const int lpg = (q.C % 4 == 0) ? q.C / 4 : 0;
    const bool vec = (lpg == 1 || lpg == 2 || lpg == 4 || lpg == 8 || lpg == 16);
    if (vec) {
        const long total = pix * lpg;
        const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
#define GO(LL) VLLM_LAUNCH((dcnv3_bwd_vec_kernel<T, LL>), grid, block, 0, st, in, off, msk, gout, gin, goff, gmsk, total, q, offset_scale)
        switch (lpg) { case 1: GO(1); break; case 2: GO(2); break; case 4: GO(4); break; case 8: GO(8); break; default: GO(16); }
#undef GO
        VLLM_CHECK_LAUNCH("dcnv3_bwd_vec_kernel");
        return VLLM_OK;
    }
    VLLM_LAUNCH((dcnv3_bwd_generic_kernel<T>), dim3((unsigned)ceil_div(pix, 256)), dim3(256), 0, st, in, off, msk, gout, gin, goff, gmsk,
                pix, q, offset_scale);
    VLLM_CHECK_LAUNCH("dcnv3_bwd_generic_kernel");
    return VLLM_OK;
}

int bwd_geo(Dcnv3Geo &q, int N, int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw)
{
    VLLM_REQUIRE(N >= 0 && H > 0 && W > 0 && G > 0 && C > 0, "dcnv3: bad tensor sizes");
    VLLM_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && dh > 0 && dw > 0, "dcnv3: bad kernel geometry");
    q = Dcnv3Geo{N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw, 0, 0};
    q.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;   // dcnv3_cuda.cu:40-45
    q.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    VLLM_REQUIRE(q.Ho > 0 && q.Wo > 0, "dcnv3: empty output (%d x %d)", q.Ho, q.Wo);
    return VLLM_OK;
}

}  // namespace
}  // namespace vllm

namespace vllm {
static int g_dcnv3_bwd_tiled = -1;
int dcnv3_bwd_tiled()
{
    if (g_dcnv3_bwd_tiled < 0) {
        const char *e = getenv("VLLM_DCNV3_BWD_TILED");
        g_dcnv3_bwd_tiled = e ? atoi(e) != 0 : 1;
    }
    return g_dcnv3_bwd_tiled;
}
int dcnv3_bwd_tiled_set(int v) { const int old = dcnv3_bwd_tiled(); g_dcnv3_bwd_tiled = v != 0; return old; }
}  // namespace vllm

using namespace vllm;

extern "C" int vllm_dcnv3_backward_f32(const float *input, const float *offset, const float *mask, const float *grad_output, int N,
                                       int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                       float offset_scale, float *grad_input, float *grad_offset, float *grad_mask,
                                       vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = bwd_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, "dcnv3_backward_f32: null pointer");
    // group channels 32 (InternImage-H style stages): the windowed kernel -- grad_input as S^T x grad_out on the fp32 MFMA, one global
    // atomic per (window pixel, channel) instead of one per (point, corner, channel) (msda_bwd_mfma.hip, template flag DCN)
    if (dcnv3_bwd_tiled() && dcnv3_bwd_mfma_takes(q) && aligned16(input) && aligned16(grad_output) && (reinterpret_cast<uintptr_t>(offset) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(grad_offset) & 7u) == 0)
        return dcnv3_bwd_mfma_launch(input, offset, mask, grad_output, q, offset_scale, grad_input, grad_offset, grad_mask, (hipStream_t)stream);
    return dcnv3_bwd_launch<float>(input, offset, mask, grad_output, q, offset_scale, grad_input, grad_offset, grad_mask, (hipStream_t)stream);
}

extern "C" int vllm_dcnv3_backward_f64(const double *input, const double *offset, const double *mask, const double *grad_output, int N,
                                       int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                       double offset_scale, double *grad_input, double *grad_offset, double *grad_mask,
                                       vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = bwd_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, "dcnv3_backward_f64: null pointer");
    return dcnv3_bwd_launch<double>(input, offset, mask, grad_output, q, offset_scale, grad_input, grad_offset, grad_mask, (hipStream_t)stream);
}

// ---- half precision (round 5): the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF (dcnv3_cuda.cu:147) with fp32 arithmetic.
// The four operands are widened into the caller's fp32 workspace, the fp32 backward above runs on them (the windowed MFMA kernel for
// group channels 32), the three gradients are rounded to half once.  (The reference accumulates grad_input with HALF atomics, one
// rounding per contribution and an order-dependent result; one rounding of the fp32 sum is the better-conditioned form of the same
// quantity, and what the parity test checks against dcnv3_core_pytorch's autograd.)
namespace vllm {
namespace {
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void widen_f16_kernel(const _Float16 *__restrict__ src, float *__restrict__ dst, long n)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const half8_t h = *reinterpret_cast<const half8_t *>(src + i);
        float4_t a = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, b = {(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
        *reinterpret_cast<float4_t *>(dst + i) = a;
        *reinterpret_cast<float4_t *>(dst + i + 4) = b;
    } else {
        for (long j = i; j < n; ++j) dst[j] = (float)src[j];
    }
}
__global__ __launch_bounds__(256) void narrow_f16_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, long n)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4_t a = *reinterpret_cast<const float4_t *>(src + i), b = *reinterpret_cast<const float4_t *>(src + i + 4);
        const half8_t h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
        *reinterpret_cast<half8_t *>(dst + i) = h;
    } else {
        for (long j = i; j < n; ++j) dst[j] = (_Float16)src[j];
    }
}
int widen(const uint16_t *src, float *dst, long n, hipStream_t st)
{
    if (n == 0) return VLLM_OK;
    VLLM_LAUNCH(widen_f16_kernel, dim3((unsigned)ceil_div(n, 2048)), dim3(256), 0, st, reinterpret_cast<const _Float16 *>(src), dst, n);
    VLLM_CHECK_LAUNCH("widen_f16_kernel");
    return VLLM_OK;
}
int narrow(const float *src, uint16_t *dst, long n, hipStream_t st)
{
    if (n == 0) return VLLM_OK;
    VLLM_LAUNCH(narrow_f16_kernel, dim3((unsigned)ceil_div(n, 2048)), dim3(256), 0, st, src, reinterpret_cast<_Float16 *>(dst), n);
    VLLM_CHECK_LAUNCH("narrow_f16_kernel");
    return VLLM_OK;
}
inline long pad4(long n) { return (n + 3) & ~3L; }   // every fp32 slab of the workspace starts 16-byte aligned
}  // namespace
}  // namespace vllm

extern "C" long vllm_dcnv3_backward_f16_workspace(int N, int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw)
{
    Dcnv3Geo q;
    if (bwd_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return -1;
    const long n_in = (long)N * H * W * G * C, n_out = (long)N * q.Ho * q.Wo * G * C, n_msk = (long)N * q.Ho * q.Wo * G * kh * kw;
    return (2 * pad4(n_in) + pad4(n_out) + 2 * pad4(2 * n_msk) + 2 * pad4(n_msk)) * (long)sizeof(float);
}

extern "C" int vllm_dcnv3_backward_f16(const uint16_t *input, const uint16_t *offset, const uint16_t *mask, const uint16_t *grad_output, int N,
                                       int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                       float offset_scale, uint16_t *grad_input, uint16_t *grad_offset, uint16_t *grad_mask,
                                       void *workspace, long workspace_bytes, vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = bwd_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, "dcnv3_backward_f16: null pointer");
    const long need = vllm_dcnv3_backward_f16_workspace(N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw);
    VLLM_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), "dcnv3_backward_f16: workspace of %ld bytes (16-byte aligned) required", need);
    //  the conversion passes move 8 halves = 16 bytes per lane on the caller's pointers
    VLLM_REQUIRE(aligned16(input) && aligned16(offset) && aligned16(mask) && aligned16(grad_output) && aligned16(grad_input) &&
                     aligned16(grad_offset) && aligned16(grad_mask),
                 "dcnv3_backward_f16: tensors must be 16-byte aligned (a view with a storage offset that is not a multiple of 8 halves: make it contiguous first)");
    const long n_in = (long)N * H * W * G * C, n_out = (long)N * q.Ho * q.Wo * G * C, n_msk = (long)N * q.Ho * q.Wo * G * kh * kw;
    hipStream_t st = (hipStream_t)stream;
    float *w = static_cast<float *>(workspace);
    float *in32 = w; w += pad4(n_in);
    float *gin32 = w; w += pad4(n_in);
    float *go32 = w; w += pad4(n_out);
    float *off32 = w; w += pad4(2 * n_msk);
    float *goff32 = w; w += pad4(2 * n_msk);
    float *msk32 = w; w += pad4(n_msk);
    float *gmsk32 = w;
    if (int e = widen(input, in32, n_in, st)) return e;
    if (int e = widen(offset, off32, 2 * n_msk, st)) return e;
    if (int e = widen(mask, msk32, n_msk, st)) return e;
    if (int e = widen(grad_output, go32, n_out, st)) return e;
    if (hipMemsetAsync(gin32, 0, (size_t)n_in * sizeof(float), st) != hipSuccess) {
        set_error("dcnv3_backward_f16: hipMemsetAsync failed");
        return VLLM_ELAUNCH;
    }
    if (int e = vllm_dcnv3_backward_f32(in32, off32, msk32, go32, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw, offset_scale, gin32, goff32, gmsk32, stream))
        return e;
    if (int e = narrow(gin32, grad_input, n_in, st)) return e;
    if (int e = narrow(goff32, grad_offset, 2 * n_msk, st)) return e;
    return narrow(gmsk32, grad_mask, n_msk, st);
}
