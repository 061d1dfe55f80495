// Region-encoder point sampling.
// Reference: point_sample = F.grid_sample(input, 2 * coords - 1, bilinear, zeros, align_corners=False)
// (visionllmv2/model/region_encoder.py:24-47) and the masked mean over a region's points (:127-141):
//     out[n, c] = sum_p valid[n, p] * sample[n, c, p] / sum_p valid[n, p]      (0 when the region has no point)
// ATen's arithmetic (grid_sampler_unnormalize / bilinear with per-corner bounds, GridSampler.h): g = 2*c - 1,
// ix = ((g + 1) * W - 1) / 2, corners floor(ix), floor(ix)+1, a corner outside the map contributes nothing.
#include "common.hpp"

namespace vllm {
namespace {

struct PsCorner {
    int x0, y0;
    float w00, w01, w10, w11;   // weights of (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1), zero where out of the map
    bool any;
};

__device__ __forceinline__ PsCorner ps_corner(float cx, float cy, int H, int W)
{
    PsCorner k;
    const float gx = 2.0f * cx - 1.0f, gy = 2.0f * cy - 1.0f;
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    // NaN / inf / far-away coordinates never reach address arithmetic
    const bool fin = ix > -2.f && iy > -2.f && ix < (float)W + 1.f && iy < (float)H + 1.f;
    const float fx = fin ? floorf(ix) : 0.f, fy = fin ? floorf(iy) : 0.f;
    k.x0 = (int)fx; k.y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const bool xl = fin && k.x0 >= 0 && k.x0 < W, xh = fin && k.x0 + 1 >= 0 && k.x0 + 1 < W;
    const bool yl = k.y0 >= 0 && k.y0 < H, yh = k.y0 + 1 >= 0 && k.y0 + 1 < H;
    k.w00 = (xl && yl) ? (1.f - tx) * (1.f - ty) : 0.f;
    k.w01 = (xh && yl) ? tx * (1.f - ty) : 0.f;
    k.w10 = (xl && yh) ? (1.f - tx) * ty : 0.f;
    k.w11 = (xh && yh) ? tx * ty : 0.f;
    k.any = (xl || xh) && (yl || yh);
    return k;
}

__device__ __forceinline__ float ps_eval(const float *__restrict__ plane, const PsCorner &k, int H, int W)
{
    if (!k.any) return 0.f;
    const int x0 = min(max(k.x0, 0), W - 1), x1 = min(max(k.x0 + 1, 0), W - 1);
    const int y0 = min(max(k.y0, 0), H - 1), y1 = min(max(k.y0 + 1, 0), H - 1);
    // a zero weight must not let a NaN at a clamped address through
    const float v00 = k.w00 != 0.f ? plane[y0 * W + x0] : 0.f, v01 = k.w01 != 0.f ? plane[y0 * W + x1] : 0.f;
    const float v10 = k.w10 != 0.f ? plane[y1 * W + x0] : 0.f, v11 = k.w11 != 0.f ? plane[y1 * W + x1] : 0.f;
    return v00 * k.w00 + v01 * k.w01 + v10 * k.w10 + v11 * k.w11;
}

// out[n, c, p]; one thread per output element, p fastest (coalesced stores, coordinates shared along c through L1/L2)
__global__ __launch_bounds__(256) void point_sample_kernel(const float *__restrict__ in, const float *__restrict__ coords,
                                                           float *__restrict__ out, long total, int C, int H, int W, int P)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % P);
    const long nc = i / P, n = nc / C;
    const float2_t xy = *reinterpret_cast<const float2_t *>(coords + (n * P + p) * 2);
    const PsCorner k = ps_corner(xy.x, xy.y, H, W);
    out[i] = ps_eval(in + nc * (long)H * W, k, H, W);
}

// out[n, c] = masked mean over the points; one block per (n, c)
__global__ __launch_bounds__(256) void point_sample_mean_kernel(const float *__restrict__ in, const float *__restrict__ coords,
                                                                const uint8_t *__restrict__ valid, float *__restrict__ out, int C,
                                                                int H, int W, int P)
{
    const long nc = blockIdx.x, n = nc / C;
    const float *plane = in + nc * (long)H * W;
    float s = 0.f, cnt = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
        if (!valid[n * P + p]) continue;
        const float2_t xy = *reinterpret_cast<const float2_t *>(coords + (n * P + p) * 2);
        s += ps_eval(plane, ps_corner(xy.x, xy.y, H, W), H, W);
        cnt += 1.f;
    }
    __shared__ float rs[4], rc[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ts = rs[0] + rs[1] + rs[2] + rs[3], tc = rc[0] + rc[1] + rc[2] + rc[3];
        out[nc] = tc > 0.f ? ts / tc : 0.f;   // (x / 0).nan_to_num() of the reference
    }
}

}  // namespace
}  // namespace vllm

using namespace vllm;

extern "C" int vllm_point_sample_f32(const float *input, const float *coords, int N, int C, int H, int W, int P, float *out,
                                     vllm_stream_t stream)
{
    VLLM_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && P >= 0, "point_sample: bad sizes");
    const long total = (long)N * C * P;
    if (total == 0) return VLLM_OK;
    VLLM_REQUIRE(input && coords && out, "point_sample: null pointer");
    VLLM_REQUIRE((reinterpret_cast<uintptr_t>(coords) & 7u) == 0, "point_sample: coords must be 8-byte aligned");
    VLLM_LAUNCH(point_sample_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, input, coords, out,
                total, C, H, W, P);
    VLLM_CHECK_LAUNCH("point_sample_kernel");
    return VLLM_OK;
}

extern "C" int vllm_point_sample_mean_f32(const float *input, const float *coords, const uint8_t *valid, int N, int C, int H,
                                          int W, int P, float *out, vllm_stream_t stream)
{
    VLLM_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && P >= 0, "point_sample_mean: bad sizes");
    if ((long)N * C == 0) return VLLM_OK;
    VLLM_REQUIRE(input && out && (P == 0 || (coords && valid)), "point_sample_mean: null pointer");
    VLLM_REQUIRE((reinterpret_cast<uintptr_t>(coords) & 7u) == 0, "point_sample_mean: coords must be 8-byte aligned");
    VLLM_LAUNCH(point_sample_mean_kernel, dim3((unsigned)((long)N * C)), dim3(256), 0, (hipStream_t)stream, input, coords, valid,
                out, C, H, W, P);
    VLLM_CHECK_LAUNCH("point_sample_mean_kernel");
    return VLLM_OK;
}
