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

// Round 5: both kernels evaluate a point's corner geometry ONCE and walk a chunk of channels with it (rounds 1-4: one thread per
// output element -- ~40 VALU of coordinate arithmetic per 4 loads and a store: 0.12 / 0.04 of the HBM roofline at the region
// encoder's shapes, profiles/r05_bench_line.json).  The zero-weight guards (a NaN at a clamped address must not get through a zero
// weight) are selects on the loaded values.
struct PsGeo {
    int o00, o01, o10, o11;     // clamped element offsets inside a plane
    float w00, w01, w10, w11;
};
__device__ __forceinline__ PsGeo ps_geo(float cx, float cy, int H, int W)
{
    const PsCorner k = ps_corner(cx, cy, H, W);
    PsGeo g;
    const int x0 = min(max(k.x0, 0), W - 1), x1 = min(max(k.x0 + 1, 0), W - 1);
    const int y0 = min(max(k.y0, 0), H - 1), y1 = min(max(k.y0 + 1, 0), H - 1);
    g.o00 = y0 * W + x0; g.o01 = y0 * W + x1; g.o10 = y1 * W + x0; g.o11 = y1 * W + x1;
    g.w00 = k.any ? k.w00 : 0.f; g.w01 = k.any ? k.w01 : 0.f; g.w10 = k.any ? k.w10 : 0.f; g.w11 = k.any ? k.w11 : 0.f;
    return g;
}
__device__ __forceinline__ float ps_eval_geo(const float *__restrict__ plane, const PsGeo &g)
{
    const float a = plane[g.o00], b = plane[g.o01], c = plane[g.o10], d = plane[g.o11];
    const float v00 = g.w00 != 0.f ? a : 0.f, v01 = g.w01 != 0.f ? b : 0.f, v10 = g.w10 != 0.f ? c : 0.f, v11 = g.w11 != 0.f ? d : 0.f;
    return v00 * g.w00 + v01 * g.w01 + v10 * g.w10 + v11 * g.w11;   // (the association of rounds 1-4: same bits)
}

constexpr int PS_CCH = 16;      // channel planes a block stages / a thread walks
constexpr int PS_LDS_MAX = 48 * 1024;   // the planes of a block in LDS when they fit (24 x 24 x 16 x 4 B = 36 KiB at the region encoder's shape)

// the block's PS_CCH planes [c][H * W] into LDS (coalesced 4-byte loads: a plane is only 16-byte aligned when H * W % 4 == 0)
__device__ __forceinline__ void ps_stage(const float *__restrict__ planes, float *lds, int n_elems)
{
    for (int i = threadIdx.x; i < n_elems; i += 256) lds[i] = planes[i];
    __syncthreads();
}
template <bool LDS>
__device__ __forceinline__ float ps_eval_any(const float *plane_g, const float *plane_l, const PsGeo &g)
{
    return ps_eval_geo(LDS ? plane_l : plane_g, g);
}

// out[n, c, p]: grid (C / PS_CCH, N).  The block's channel planes are staged in LDS once (LDS = true; 2.3 KB each at 24 x 24) and every
// thread walks points p = tid, tid + 256, ...: corner geometry once per point, then per channel four LDS reads, four multiply-adds and
// one store (p fastest: 256 contiguous bytes per wave and channel).  LDS = false: the same walk straight from global memory (maps too
// large to stage).
template <bool LDS>
__global__ __launch_bounds__(256) void point_sample_kernel(const float *__restrict__ in, const float *__restrict__ coords,
                                                           float *__restrict__ out, int C, int H, int W, int P)
{
    extern __shared__ __attribute__((aligned(16))) float ps_lds[];
    const int c0 = blockIdx.x * PS_CCH;
    const long n = blockIdx.y;
    const int HW = H * W, nc = min(PS_CCH, C - c0);
    const float *planes = in + (n * C + c0) * (long)HW;
    if (LDS) ps_stage(planes, ps_lds, nc * HW);
    for (int p = threadIdx.x; p < P; p += 256) {
        const float2_t xy = *reinterpret_cast<const float2_t *>(coords + (n * P + p) * 2);
        const PsGeo g = ps_geo(xy.x, xy.y, H, W);
        float *o = out + (n * C + c0) * (long)P + p;
#pragma unroll 4
        for (int c = 0; c < nc; ++c) o[(long)c * P] = ps_eval_any<LDS>(planes + (long)c * HW, ps_lds + c * HW, g);
    }
}

// out[n, c] = masked mean over the points, the walk of point_sample_kernel<false> with PS_CCH partial sums per thread (maps too large
// for the pixel-weight form below); wave reduction by xor shuffles, the four waves through LDS, in a fixed order
__global__ __launch_bounds__(256) void point_sample_mean_kernel(const float *__restrict__ in, const float *__restrict__ coords,
                                                                const uint8_t *__restrict__ valid, float *__restrict__ out, int C,
                                                                int H, int W, int P)
{
    const int c0 = blockIdx.x * PS_CCH;
    const long n = blockIdx.y;
    const int HW = H * W, nc = min(PS_CCH, C - c0);
    const float *planes = in + (n * C + c0) * (long)HW;
    float s[PS_CCH];
#pragma unroll
    for (int c = 0; c < PS_CCH; ++c) s[c] = 0.f;
    float cnt = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
        if (!valid[n * P + p]) continue;
        const float2_t xy = *reinterpret_cast<const float2_t *>(coords + (n * P + p) * 2);
        const PsGeo g = ps_geo(xy.x, xy.y, H, W);
        cnt += 1.f;
#pragma unroll
        for (int c = 0; c < PS_CCH; ++c)
            if (c < nc) s[c] += ps_eval_geo(planes + (long)c * HW, g);
    }
    __shared__ float rs[4][PS_CCH], rc[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
#pragma unroll
        for (int c = 0; c < PS_CCH; ++c) s[c] += __shfl_xor(s[c], o);
    }
    if ((threadIdx.x & 63) == 0) {
        rc[threadIdx.x >> 6] = cnt;
#pragma unroll
        for (int c = 0; c < PS_CCH; ++c) rs[threadIdx.x >> 6][c] = s[c];
    }
    __syncthreads();
    if ((int)threadIdx.x < nc) {
        const int c = threadIdx.x;
        const float ts = (rs[0][c] + rs[1][c]) + (rs[2][c] + rs[3][c]), tc = (rc[0] + rc[1]) + (rc[2] + rc[3]);
        out[n * C + c0 + c] = tc > 0.f ? ts / tc : 0.f;   // (x / 0).nan_to_num() of the reference
    }
}

// Round 6: the masked mean as a PIXEL-weight product.  The mean over a region's points of a bilinear sample is linear in the map:
//     out[n, c] = (1 / cnt) sum_pix A[n, pix] in[n, c, pix],      A[n, pix] = sum over the valid points of the corner weight they put on pix
// and A does not depend on the channel: 2304 points x 4 corners are scattered ONCE per block into H W accumulators, then every
// channel is ONE dot product with its plane -- each feature value is read exactly once, with one multiply-add, instead of four
// LDS reads + four multiply-adds per (point, channel) (round 5: 103 us = 0.13 of the HBM roof at 16 regions x 3072 channels x 24 x 24).
// Deterministic: the scatter adds 2^40-scaled weights as 64-bit INTEGERS (LDS atomics commute exactly; a weight is in [0, 1], a
// pixel collects at most P of them: < 2^52), the dot product has a fixed order (lane-strided partial sums, one xor tree).  The
// result differs from the reference's summation order (points first) by fp32 rounding only; the weights themselves are
// ps_geo's, bit for bit.  grid (ceil(C / PM_CPB), N); LDS: H W x (8 + 4) bytes.
constexpr int PM_CPB = 96;      // channels per block (24 per wave)
template <bool V4>
__global__ __launch_bounds__(256) void point_sample_mean_pix_kernel(const float *__restrict__ in, const float *__restrict__ coords,
                                                                    const uint8_t *__restrict__ valid, float *__restrict__ out, int C,
                                                                    int H, int W, int P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long pm_acc[];   // [HW] fixed-point sums, then [HW] floats behind them
    __shared__ int s_cnt;
    const int HW = H * W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *A = reinterpret_cast<float *>(pm_acc + ((HW + 1) & ~1));
    const int c0 = blockIdx.x * PM_CPB;
    const long n = blockIdx.y;
    for (int i = tid; i < HW; i += 256) pm_acc[i] = 0ull;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int p = tid; p < P; p += 256) {
        if (!valid[n * P + p]) continue;
        const float2_t xy = *reinterpret_cast<const float2_t *>(coords + (n * P + p) * 2);
        const PsGeo g = ps_geo(xy.x, xy.y, H, W);
        ++cnt;
        constexpr float SC = 1099511627776.0f;   // 2^40
        if (g.w00 != 0.f) atomicAdd(pm_acc + g.o00, (unsigned long long)(g.w00 * SC));
        if (g.w01 != 0.f) atomicAdd(pm_acc + g.o01, (unsigned long long)(g.w01 * SC));
        if (g.w10 != 0.f) atomicAdd(pm_acc + g.o10, (unsigned long long)(g.w10 * SC));
        if (g.w11 != 0.f) atomicAdd(pm_acc + g.o11, (unsigned long long)(g.w11 * SC));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    for (int i = tid; i < HW; i += 256) A[i] = (float)((double)pm_acc[i] * (1.0 / 1099511627776.0));
    __syncthreads();
    const int tc = s_cnt;
    const int nc = min(PM_CPB, C - c0);
    // a wave takes channels wave, wave + 4, ...; two at a time (their loads in flight together)
    for (int cc = wave; cc < nc; cc += 8) {
        const int ca = cc, cb = cc + 4;
        const bool hb = cb < nc;
        const float *pa = in + ((n * C + c0 + ca) * (long)HW), *pb = in + ((n * C + c0 + (hb ? cb : ca)) * (long)HW);
        float sa = 0.f, sb = 0.f;
        if (V4) {
            for (int i = lane * 4; i < HW; i += 256) {
                const float4_t wv = *reinterpret_cast<const float4_t *>(A + i);
                const float4_t xa = *reinterpret_cast<const float4_t *>(pa + i), xb = *reinterpret_cast<const float4_t *>(pb + i);
                // (a non-finite value at a pixel no valid point touches must not get through its zero weight: selects, as in ps_eval_geo)
                float ta[4], tb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { ta[k] = wv[k] != 0.f ? wv[k] * xa[k] : 0.f; tb[k] = wv[k] != 0.f ? wv[k] * xb[k] : 0.f; }
                sa += (ta[0] + ta[1]) + (ta[2] + ta[3]);
                sb += (tb[0] + tb[1]) + (tb[2] + tb[3]);
            }
        } else {
            for (int i = lane; i < HW; i += 64) {
                const float wv = A[i];
                sa += wv != 0.f ? wv * pa[i] : 0.f;
                sb += wv != 0.f ? wv * pb[i] : 0.f;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        if (lane == 0) {
            // (ts / tc as the walk above: the reference's (x / 0).nan_to_num() is 0)
            out[n * C + c0 + ca] = tc > 0 ? sa / (float)tc : 0.f;
            if (hb) out[n * C + c0 + cb] = tc > 0 ? sb / (float)tc : 0.f;
        }
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
    VLLM_REQUIRE(N <= 65535 && (long)H * W < (1L << 24), "point_sample: too many regions / too large a map for one launch");
    const size_t lds = (size_t)PS_CCH * H * W * sizeof(float);
    const dim3 grid((unsigned)ceil_div(C, PS_CCH), (unsigned)N);
    if (lds <= (size_t)PS_LDS_MAX) VLLM_LAUNCH((point_sample_kernel<true>), grid, dim3(256), lds, (hipStream_t)stream, input, coords, out, C, H, W, P);
    else VLLM_LAUNCH((point_sample_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, input, coords, out, C, H, W, P);
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
    VLLM_REQUIRE(N <= 65535 && (long)H * W < (1L << 24), "point_sample_mean: too many regions / too large a map for one launch");
    const long HW = (long)H * W;
    const size_t lds = (size_t)((HW + 1) & ~1L) * 8 + (size_t)HW * 4;
    hipStream_t st = (hipStream_t)stream;
    if (lds <= (size_t)PS_LDS_MAX) {   // the pixel-weight form (round 6)
        const dim3 grid((unsigned)ceil_div(C, PM_CPB), (unsigned)N);
        if (HW % 4 == 0 && aligned16(input)) VLLM_LAUNCH((point_sample_mean_pix_kernel<true>), grid, dim3(256), lds, st, input, coords, valid, out, C, H, W, P);
        else VLLM_LAUNCH((point_sample_mean_pix_kernel<false>), grid, dim3(256), lds, st, input, coords, valid, out, C, H, W, P);
    } else {
        const dim3 grid((unsigned)ceil_div(C, PS_CCH), (unsigned)N);
        VLLM_LAUNCH(point_sample_mean_kernel, grid, dim3(256), 0, st, input, coords, valid, out, C, H, W, P);
    }
    VLLM_CHECK_LAUNCH("point_sample_mean_kernel");
    return VLLM_OK;
}
