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



}  // namespace vllm
