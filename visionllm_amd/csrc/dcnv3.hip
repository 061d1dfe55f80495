// DCNv3 forward (deformable convolution v3 of the InternImage det backbone) -- SURVEY section 8 row f3.
// Reference: visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:31-84 (bilinear sample of a [H, W, G, C] image
// with zero padding), :217-278 (forward kernel: reference point p0 of an output pixel, kernel_w-outer / kernel_h-inner
// point order, acceptance test), dcnv3_cuda.cu:40-45 (output geometry).
//
// Same access pattern as the MSDA gather kernel, one "level": a lane owns VEC consecutive channels of one
// (batch, output pixel, group); the C / VEC lanes of a group sit next to each other, so a corner is one contiguous
// read of the group's channels and the group's offsets / mask values are broadcast loads.  Corner loads are issued from
// clamped addresses and selects decide what contributes (a NaN at a clamped address cannot leak).
#include "common.hpp"
#include "dcnv3_geo.hpp"

namespace vllm {

bool dcnv3_tiled_ok(const Dcnv3Geo &q, const float *in, const float *off, const float *msk, const float *out);   // dcnv3_tiled.hip
int dcnv3_tiled_launch(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out,
                       hipStream_t st);
bool dcnv3_pipe_ok(const Dcnv3Geo &q);                                                                           // dcnv3_pipe.hip
int dcnv3_pipe_launch(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out, int prof,
                      hipStream_t st);
int dcnv3_tiled_enabled();   // runtime.cpp: 0 gather kernel, 1 pipelined tiled kernel (2: + phase clock), 3 two-block tiled kernel (4: + phase clock)

namespace {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Opmath { typedef T type; };
template <typename T> __device__ __forceinline__ T floor_t(T x);
template <> __device__ __forceinline__ float floor_t<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double floor_t<double>(double x) { return floor(x); }

// K3 = true: 3x3 kernel, the 9-point loop is fully unrolled (36 independent corner loads per lane for the scheduler).
// TIO: storage type of the four tensors (float, double, or -- round 5 -- _Float16: the reference dispatches
// AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69, with opmath_t = float: half operands, fp32 arithmetic, one rounding of
// the output); T: the arithmetic type.
template <typename TIO, typename T, int VEC, bool K3>
__global__ __launch_bounds__(256) void dcnv3_fwd_kernel(const TIO *__restrict__ in, const TIO *__restrict__ off,
                                                        const TIO *__restrict__ msk, TIO *__restrict__ out, long total,
                                                        Dcnv3Geo q, T offset_scale)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cpl = q.C / VEC;                       // lanes per (pixel, group)
    const int cc = (int)(idx % cpl);
    const long sidx = idx / cpl;                     // ((b * Ho + y) * Wo + x) * G + g
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
    const TIO *im = in + b * q.H * hs + (long)g * q.C + cc * VEC;
    const TIO *op = off + sidx * K * 2;
    const TIO *mp = msk + sidx * K;
    T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    auto point = [&](int i, int j) {
            const T off_w = (T)op[0], off_h = (T)op[1], wgt = (T)mp[0];
            op += 2; mp += 1;
            const T loc_w = dcn_loc<T>(p0_w_, (T)(i * q.dw), off_w, offset_scale);
            const T loc_h = dcn_loc<T>(p0_h_, (T)(j * q.dh), off_h, offset_scale);
            const bool ok = loc_h > (T)-1 && loc_w > (T)-1 && loc_h < (T)q.H && loc_w < (T)q.W;
            // a rejected location (possibly NaN / inf) never reaches the address arithmetic
            const int h_low = ok ? (int)floor_t<T>(loc_h) : 0, w_low = ok ? (int)floor_t<T>(loc_w) : 0;
            const T lh = loc_h - (T)h_low, lw = loc_w - (T)w_low, hh = (T)1 - lh, hw = (T)1 - lw;
            const bool u0 = ok && h_low >= 0, u1 = ok && h_low + 1 <= q.H - 1, l0 = w_low >= 0, l1 = w_low + 1 <= q.W - 1;
            const int y0 = min(max(h_low, 0), q.H - 1), y1 = min(max(h_low + 1, 0), q.H - 1);
            const int x0 = min(max(w_low, 0), q.W - 1), x1 = min(max(w_low + 1, 0), q.W - 1);
            const TIO *c1 = im + y0 * hs + x0 * ws, *c2 = im + y0 * hs + x1 * ws, *c3 = im + y1 * hs + x0 * ws,
                      *c4 = im + y1 * hs + x1 * ws;
            const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            T v1[VEC], v2[VEC], v3[VEC], v4[VEC];
            if constexpr (VEC == 4 && sizeof(TIO) == 4) {
                *reinterpret_cast<float4_t *>(v1) = *reinterpret_cast<const float4_t *>(c1);
                *reinterpret_cast<float4_t *>(v2) = *reinterpret_cast<const float4_t *>(c2);
                *reinterpret_cast<float4_t *>(v3) = *reinterpret_cast<const float4_t *>(c3);
                *reinterpret_cast<float4_t *>(v4) = *reinterpret_cast<const float4_t *>(c4);
            } else if constexpr (VEC == 4 && sizeof(TIO) == 2) {   // four halves = one 8-byte load per corner
                const half4_t h1 = *reinterpret_cast<const half4_t *>(c1), h2 = *reinterpret_cast<const half4_t *>(c2);
                const half4_t h3 = *reinterpret_cast<const half4_t *>(c3), h4 = *reinterpret_cast<const half4_t *>(c4);
#pragma unroll
                for (int v = 0; v < 4; ++v) { v1[v] = (T)h1[v]; v2[v] = (T)h2[v]; v3[v] = (T)h3[v]; v4[v] = (T)h4[v]; }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) { v1[v] = (T)c1[v]; v2[v] = (T)c2[v]; v3[v] = (T)c3[v]; v4[v] = (T)c4[v]; }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T s = w1 * ((u0 && l0) ? v1[v] : (T)0) + w2 * ((u0 && l1) ? v2[v] : (T)0) +
                            w3 * ((u1 && l0) ? v3[v] : (T)0) + w4 * ((u1 && l1) ? v4[v] : (T)0);
                acc[v] += ok ? s * wgt : (T)0;
            }
            };
    if constexpr (K3) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) point(i, j);
    } else {
        for (int i = 0; i < q.kw; ++i)
            for (int j = 0; j < q.kh; ++j) point(i, j);
    }
    TIO *o = out + sidx * q.C + cc * VEC;
    if constexpr (VEC == 4 && sizeof(TIO) == 4) {
        *reinterpret_cast<float4_t *>(o) = *reinterpret_cast<const float4_t *>(acc);
    } else if constexpr (VEC == 4 && sizeof(TIO) == 2) {
        const half4_t h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};   // round to nearest even, once
        *reinterpret_cast<half4_t *>(o) = h;
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = (TIO)acc[v];
    }
}

template <typename TIO, typename T>
int dcnv3_launch(const TIO *in, const TIO *off, const TIO *msk, Dcnv3Geo q, T offset_scale, TIO *out, hipStream_t st)
{
    const long pix = (long)q.N * q.Ho * q.Wo * q.G;
    if (pix == 0 || q.C == 0) return VLLM_OK;
    const bool vec = sizeof(TIO) <= 4 && q.C % 4 == 0 && aligned16(in) && aligned16(out);
    const long total = pix * (vec ? q.C / 4 : q.C);
    VLLM_REQUIRE(total < (1L << 40), "dcnv3: too many output elements");
    const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
    const bool k3 = q.kh == 3 && q.kw == 3;
    if (vec && k3) VLLM_LAUNCH((dcnv3_fwd_kernel<TIO, T, 4, true>), grid, block, 0, st, in, off, msk, out, total, q, offset_scale);
    else if (vec) VLLM_LAUNCH((dcnv3_fwd_kernel<TIO, T, 4, false>), grid, block, 0, st, in, off, msk, out, total, q, offset_scale);
    else if (k3) VLLM_LAUNCH((dcnv3_fwd_kernel<TIO, T, 1, true>), grid, block, 0, st, in, off, msk, out, total, q, offset_scale);
    else VLLM_LAUNCH((dcnv3_fwd_kernel<TIO, T, 1, false>), grid, block, 0, st, in, off, msk, out, total, q, offset_scale);
    VLLM_CHECK_LAUNCH("dcnv3_fwd_kernel");
    return VLLM_OK;
}

int make_geo(Dcnv3Geo &q, int N, int H, int W, int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw)
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

using namespace vllm;

extern "C" int vllm_dcnv3_forward_f32(const float *input, const float *offset, const float *mask, int N, int H, int W,
                                      int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                      float offset_scale, float *out, vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = make_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && out, "dcnv3_forward_f32: null pointer");
    if (dcnv3_tiled_ok(q, input, offset, mask, out)) {   // group channels 16 / 32, <= 9 points: the LDS-tiled kernels
        const int mode = dcnv3_tiled_enabled();
        if (mode <= 2 && dcnv3_pipe_ok(q)) return dcnv3_pipe_launch(input, offset, mask, q, offset_scale, out, mode == 2, (hipStream_t)stream);
        return dcnv3_tiled_launch(input, offset, mask, q, offset_scale, out, (hipStream_t)stream);
    }
    return dcnv3_launch<float, float>(input, offset, mask, q, offset_scale, out, (hipStream_t)stream);
}

extern "C" int vllm_dcnv3_forward_f64(const double *input, const double *offset, const double *mask, int N, int H, int W,
                                      int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                      double offset_scale, double *out, vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = make_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && out, "dcnv3_forward_f64: null pointer");
    return dcnv3_launch<double, double>(input, offset, mask, q, offset_scale, out, (hipStream_t)stream);
}

// Half precision (round 5; the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69): IEEE binary16 tensors as
// uint16_t bit patterns, fp32 arithmetic (opmath_t = float), the output rounded once -- the result of running the fp32 operator on
// the widened operands and rounding it, which is what the parity test checks bit for bit.
extern "C" int vllm_dcnv3_forward_f16(const uint16_t *input, const uint16_t *offset, const uint16_t *mask, int N, int H, int W,
                                      int G, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                      float offset_scale, uint16_t *out, vllm_stream_t stream)
{
    Dcnv3Geo q;
    if (int e = make_geo(q, N, H, W, G, C, kh, kw, sh, sw, ph, pw, dh, dw)) return e;
    if (N == 0) return VLLM_OK;
    VLLM_REQUIRE(input && offset && mask && out, "dcnv3_forward_f16: null pointer");
    return dcnv3_launch<_Float16, float>(reinterpret_cast<const _Float16 *>(input), reinterpret_cast<const _Float16 *>(offset),
                                         reinterpret_cast<const _Float16 *>(mask), q, offset_scale, reinterpret_cast<_Float16 *>(out),
                                         (hipStream_t)stream);
}
