// Helpers shared by the pyramid-item MSDA forward kernels (msda_tiled6.hip, msda_tiled7.hip).
#pragma once
#include "common.hpp"

namespace vllm {
namespace {

constexpr int T6_ZPX = 48;               // zero strip at the bottom of LDS [pixels]; a window's pitch must stay <= ZPX - 2
constexpr int T6_BIG = 0x3fffffff;
constexpr int T6_SLACK = 8;              // windows are padded to 8 pixels (the last DMA instruction writes whole groups)

__device__ __attribute__((aligned(128))) float g_t6_zero_px[32];   // zero-initialised: DMA source of out-of-image pixels

template <int K>
__device__ __forceinline__ int qbi(int x)   // value of lane K of this lane's quad
{
    return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, true);
}
template <int K>
__device__ __forceinline__ float qbf(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, true));
}
// acc += (w of quad lane K) * v in ONE VALU instruction
template <int K>
__device__ __forceinline__ void fmac_q(float &acc, float w, float v)
{
    static_assert(K >= 0 && K < 4, "quad lane");
    if constexpr (K == 0) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
    if constexpr (K == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v));
}
__device__ __forceinline__ float2_t t6_fma2(float w, float2_t v, float2_t a)
{
    return __builtin_elementwise_fma((float2_t){w, w}, v, a);
}
template <int CTRL>
__device__ __forceinline__ int dpp_min(int v)   // min(v, v of the lane CTRL selects); row_ror keeps lane & 3
{
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ int sel4(int k, int a0, int a1, int a2, int a3)
{
    return k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
}
__device__ __forceinline__ unsigned lds_addr(const void *p)   // LDS byte address of a __shared__ object
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

// 16 bytes of fp32 channels from a value row of either storage type (cold path)
__device__ __forceinline__ float4_t load4(const float *p) { return *reinterpret_cast<const float4_t *>(p); }
__device__ __forceinline__ float4_t load4(const uint16_t *p)
{
    const uint2_t r = *reinterpret_cast<const uint2_t *>(p);
    return (float4_t){bf16lo_to_f32(r.x), bf16hi_to_f32(r.x), bf16lo_to_f32(r.y), bf16hi_to_f32(r.y)};
}
__device__ __forceinline__ void store4(float *p, float4_t v) { *reinterpret_cast<float4_t *>(p) = v; }
__device__ __forceinline__ void store4(uint16_t *p, float4_t v)
{
    *reinterpret_cast<uint2_t *>(p) = (uint2_t){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
}

}  // namespace
}  // namespace vllm
