// Shared by the attention kernels (attn.hip; tools/experiments/attn2.hip: the round-3 hand-placed schedule, same speed).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace vllm {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int ATT_THREADS = 256;
constexpr int QBLK = 128;   // query rows per block
constexpr int KVBLK = 64;   // keys per tile

template <int D> __device__ __forceinline__ int swz_k(int row) { return D == 64 ? ((row >> 1) & 7) : (row & 15); }
template <int D> __device__ __forceinline__ int swz_v(int row) { return D == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }

// Combine the two key halves of a query (lanes l and l^32) with ONE v_permlane32_swap instead of a ds_bpermute round
// trip: swapping x with itself leaves {x_lo, x_lo} in one register and {x_hi, x_hi} in the other.
// (Toolchain note, round 3: in SMALL kernels this compiler folds the builtin's two results into one register -- r[0] + r[1]
//  becomes v + v, fmaxf(r[0], r[1]) becomes v: profiles/r03_permlane_swap_codegen.txt.  In the attention kernels both results
//  are read (tests/test_capi.py::test_attention_lane_swaps_read_both_results scans the disassembly after every build of the
//  test suite); gemm256p.hip, where the folding did happen, does its exchanges in inline assembly.)
__device__ __forceinline__ float halves_max(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The 16-bit element type of q / k / v / P / out is a template flag: bf16 (the vision tower's dtype) or IEEE half (the reference's
// FlashAttention accepts both, flash_attention.py:39-41).  Same instruction count either way: 32x32x16 MFMA, packed
// round-to-nearest-even conversion, dot2 against (1, 1) for the row sums; the LDS layouts and the transpose reads only see 16-bit words.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ f32x16_t mfma16(bf16x8_t a, bf16x8_t b, f32x16_t c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ uint32_t pack16x2(float a, float b)
{
    if constexpr (F16) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));   // round to nearest even
    } else return pack_bf16x2(a, b);
}
template <bool F16>
__device__ __forceinline__ float dot2_ones(uint32_t w, float acc)
{
    if constexpr (F16) {
        const f16x2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), ones, acc, false);
    } else {
        const bf16x2v_t ones = {(__bf16)1.0f, (__bf16)1.0f};
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, w), ones, acc, false);
    }
}

}  // namespace vllm
