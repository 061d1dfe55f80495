// Shared fp32 epilogue of the bf16 GEMM kernels: a lane owns 4 consecutive output features n..n+3 of token m.
#pragma once
#include "kernels.hpp"

namespace vllm {

// The epilogue of a 256x256 tile evaluates 64K activations per block: libm erff (~40 instructions) and an IEEE
// division made it ~6 us of pure VALU per tile.  These forms are 1 v_exp + 1 v_rcp + a handful of FMAs and are exact
// to well below bf16 resolution (erf: Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7; v_rcp_f32: 1 ulp).
__device__ __forceinline__ float fast_erf(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float quick_gelu(float x)
{
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}

struct EpiCols {   // per-n constants, loaded once per 16-column tile
    float bia[4], scl[4];
};

template <int EPI>
__device__ __forceinline__ EpiCols epi_cols(const GemmArgs &a, int n)
{
    EpiCols c;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c.bia[r] = 0.f; c.scl[r] = 1.f; }
    const uint16_t *bias = a.bias;
    if (EPI == EPI_MSDA && n >= a.nsplit) bias = a.bias2 ? a.bias2 - a.nsplit : nullptr;
    if (bias) {
        const uint2_t b = *reinterpret_cast<const uint2_t *>(bias + n);
        c.bia[0] = bf16lo_to_f32(b.x); c.bia[1] = bf16hi_to_f32(b.x); c.bia[2] = bf16lo_to_f32(b.y); c.bia[3] = bf16hi_to_f32(b.y);
    }
    if (EPI == EPI_RESIDUAL && a.scale) {
        const uint2_t s = *reinterpret_cast<const uint2_t *>(a.scale + n);
        c.scl[0] = bf16lo_to_f32(s.x); c.scl[1] = bf16hi_to_f32(s.x); c.scl[2] = bf16lo_to_f32(s.y); c.scl[3] = bf16hi_to_f32(s.y);
    }
    return c;
}

// Final fp32 values of features n .. n+3 of token m (bias, activation, LayerScale + residual, position embedding).
template <int EPI, typename ACC4>
__device__ __forceinline__ void epi_value(const GemmArgs &a, int m, int n, const ACC4 &acc, const EpiCols &c, float (&v)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + c.bia[r];
    if (EPI == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    } else if (EPI == EPI_QUICK_GELU) {
        // x * sigmoid(1.702 x), x = acc + bias: the exponent -1.702 log2(e) (acc + bias) is ONE fma of the accumulator (the bias
        // times the constant is per column) instead of add + multiply -- 5.5 instead of 6.5 VALU per element of a tile whose
        // epilogue nothing overlaps
        constexpr float kq = -1.702f * 1.4426950408889634f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[r], kq, c.bia[r] * kq)));
    } else if (EPI == EPI_RESIDUAL && a.res_init) {
        // the residual went in as the accumulators' initial value (gemm256), divided by the LayerScale where there is one:
        // (res / ls + X W^T + bias) * ls = res + (X W^T + bias) * ls   (scl = 1 without LayerScale)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= c.scl[r];
    } else if (EPI == EPI_RESIDUAL) {
        const uint2_t rr = *reinterpret_cast<const uint2_t *>(a.res + (size_t)m * a.ldr + n);
        v[0] = bf16lo_to_f32(rr.x) + v[0] * c.scl[0]; v[1] = bf16hi_to_f32(rr.x) + v[1] * c.scl[1];
        v[2] = bf16lo_to_f32(rr.y) + v[2] * c.scl[2]; v[3] = bf16hi_to_f32(rr.y) + v[3] * c.scl[3];
    } else if (EPI == EPI_EMBED) {
        const int img = m / a.P, p = m - img * a.P;
        const uint2_t pp = *reinterpret_cast<const uint2_t *>(a.res + (size_t)(1 + p) * a.ldr + n);
        v[0] += bf16lo_to_f32(pp.x); v[1] += bf16hi_to_f32(pp.x); v[2] += bf16lo_to_f32(pp.y); v[3] += bf16hi_to_f32(pp.y);
    }
}

// The same for the consumer of a folded norm (bias / GELU / quick-GELU): x = r_m acc - r_m mean_m colsum_n + bias'_n as two
// fmas per element, rn = {r_m, -r_m mean_m}.
template <int EPI, typename ACC4, typename F4>
__device__ __forceinline__ void epi_value_folded(const ACC4 &acc, float rn_x, float rn_y, const F4 &colsum, const EpiCols &c, float (&v)[4])
{
    static_assert(EPI == EPI_BIAS || EPI == EPI_GELU || EPI == EPI_QUICK_GELU, "a folded norm feeds a bias / GELU / quick-GELU epilogue");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = fmaf(rn_x, acc[r], fmaf(rn_y, colsum[r], c.bia[r]));
        if (EPI == EPI_GELU) {
            v[r] = gelu_erf(x);
        } else if (EPI == EPI_QUICK_GELU) {
            constexpr float kq = -1.702f * 1.4426950408889634f;
            v[r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * kq));
        } else {
            v[r] = x;
        }
    }
}

// Output row of GEMM row m (the patch-embedding epilogue scatters past the CLS slot of every image).
template <int EPI>
__device__ __forceinline__ size_t epi_out_row(const GemmArgs &a, int m)
{
    if (EPI == EPI_EMBED) {
        const int img = m / a.P, p = m - img * a.P;
        return (size_t)img * (a.P + 1) + 1 + p;
    }
    return (size_t)m;
}

template <int EPI, typename ACC4>
__device__ __forceinline__ void epi_store(const GemmArgs &a, int m, int n, const ACC4 &acc, const EpiCols &c)
{
    float v[4];
    epi_value<EPI>(a, m, n, acc, c, v);
    const size_t orow = epi_out_row<EPI>(a, m);
    if (EPI == EPI_F32) {   // fp32 result (Y is float*, ldy in floats); `res` = optional uint8 row mask -> zero rows
        const uint8_t *mask = reinterpret_cast<const uint8_t *>(a.res);
        const bool dead = mask && mask[m] != 0;
        const float4_t o4 = {dead ? 0.f : v[0], dead ? 0.f : v[1], dead ? 0.f : v[2], dead ? 0.f : v[3]};
        *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(a.Y) + orow * a.ldy + n) = o4;
        return;
    }
    uint2_t o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2_t *>(a.Y + orow * a.ldy + n) = o;
}

}  // namespace vllm
