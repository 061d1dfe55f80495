// Shared by the attention kernels (attn.hip; tools/experiments/attn2.hip: the round-3 hand-placed schedule, same speed;
// tools/experiments/attn64.hip: round 6's 64-rows-per-wave body, slower).
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

// K/V staging.  The per-lane part of every source address (row-in-tile * token stride + swizzled 16-byte chunk) does not
// change from tile to tile: it is computed ONCE (kv_lane_offsets) and each tile's LDS-DMA is then
// (uniform tile base in SGPRs) + (that 32-bit lane offset) -> the saddr form of global_load_lds with a uniform LDS
// destination.  Left per tile, the 64-bit row multiply / clamp / readfirstlane chain was ~12 VALU per tile on a kernel
// whose VALU issue is the bound.  Only the last tile of a ragged S clamps rows (keys >= S re-read key S-1; masked later).
template <int D, int WPB = 4> struct KvStage {   // WPB = waves per block sharing the staging work
    static constexpr int CPR = D / 8;           // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;        // rows per wave instruction (1 KiB)
    static constexpr int NI = KVBLK / RPI / WPB;  // instructions per wave
};

template <int D, bool ISV, int WPB = 4>
__device__ __forceinline__ void kv_lane_offsets(int ts, int wave, int lane, uint32_t (&vo)[KvStage<D, WPB>::NI])
{
    typedef KvStage<D, WPB> G;
#pragma unroll
    for (int s = 0; s < G::NI; ++s) {
        const int r = (wave * G::NI + s) * G::RPI + lane / G::CPR;
        const int c = (lane % G::CPR) ^ (ISV ? swz_v<D>(r) : swz_k<D>(r));
        vo[s] = (uint32_t)(r * ts + c * 8) * 2u;
    }
}

template <int D, bool ISV, bool RAGGED = true, int WPB = 4>
__device__ __forceinline__ void stage_kv(const uint16_t *__restrict__ base, int ts, int k0, int S, char *lds_tile,
                                         int wave, int lane, const uint32_t (&vo)[KvStage<D, WPB>::NI])
{
    typedef KvStage<D, WPB> G;
    uint32_t off[G::NI];
#pragma unroll
    for (int s = 0; s < G::NI; ++s) off[s] = vo[s];
    if (RAGGED && k0 + KVBLK > S) {   // block-uniform: ragged last tile, rows past the last key re-read key S-1
#pragma unroll
        for (int s = 0; s < G::NI; ++s) {
            const int r = (wave * G::NI + s) * G::RPI + lane / G::CPR;
            const int c = (lane % G::CPR) ^ (ISV ? swz_v<D>(r) : swz_k<D>(r));   // the LDS image keeps row r's swizzle
            const int rr = k0 + r < S ? r : S - 1 - k0;
            off[s] = (uint32_t)(rr * ts + c * 8) * 2u;
        }
    }
    // ONE load site per instruction: (uniform tile base) + (32-bit lane offset) selects the saddr form, uniform LDS address.
    // The empty asm keeps the zero-extension of the offset from being hoisted out of the tile loop as a 64-bit register
    // pair (which turns every DMA back into a 64-bit VALU add + vaddr form).
    const char *tile = reinterpret_cast<const char *>(base + (long)k0 * ts);
#pragma unroll
    for (int s = 0; s < G::NI; ++s) asm volatile("" : "+v"(off[s]));
#pragma unroll
    for (int s = 0; s < G::NI; ++s)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tile + off[s]),
                                         (__attribute__((address_space(3))) void *)(lds_tile + (wave * G::NI + s) * 1024), 16,
                                         0, 0);
}

// ---- the CLS "+1" (round 6) ------------------------------------------------------------------------------------------------
// A ViT tile is S = 1 + n^2 tokens (modeling_intern_vit.py:85-89 concatenates the class token in front): 577 = 9 * 64 + 1,
// 1025 = 16 * 64 + 1.  Tiled as it stands, the single extra token costs a tenth key tile and a fifth / ninth query block per
// (tile, head) -- ~15 % of the MFMA work at ViT-L.  When (S - 1) % 64 == 0 the launcher takes token 0 out of the tilings:
//   * as a KEY it is the initial state of every query's online softmax (m = q.k0 * c, l = 1, O = v0: one dot product and D
//     loads per lane) -- the tile loop then walks keys 1 .. S-1 in exact 64-key tiles;
//   * as a QUERY it moves into the spare wave of the last query block where the body rows leave one (576 rows = 4.5 blocks of
//     four waves: the MFMA path with one live row, at no cost in block slots).  Where they do not (1024 rows = 8 blocks exactly)
//     it stays where it was -- row 0 of the ordinary query tiling, 9 blocks -- and only the KEY side is split.
//   Measured in round 6 (profiles/r06_attn_cls_split.txt; 40 tiles): d 64 / S 577: 77.6-78.5 us against 80.8-83.6 for the plain
//   tiling; d 128 / S 1025: keys only 641 us, plain 650, keys + a VALU block per pair for the class row 654-674 (the block's
//   serial load loops hold a block slot for ~10 us each); 192-row blocks of six waves for the 576 body rows (3 blocks per pair
//   exactly, but 12 instead of 16 resident waves per CU): 123 us.  The losing forms are not in the library
//   (tools/experiments/attn64.hip keeps the VALU class row next to the 64-rows-per-wave body that used it).
// Results differ from the plain tiling by summation order only (P is rounded to 16 bits before P V on both paths).
template <bool F16> __device__ __forceinline__ float cvt16(uint32_t h)
{
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
    else return __uint_as_float(h << 16);
}
template <bool F16> __device__ __forceinline__ float dot2_acc(uint32_t x, uint32_t y, float acc)
{
    if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, x), __builtin_bit_cast(f16x2_t, y), acc, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, x), __builtin_bit_cast(bf16x2v_t, y), acc, false);
}
}  // namespace vllm
