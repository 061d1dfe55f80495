// Shared by msda.hip and msda_tiled.hip: the ONE definition of the integer part of a sampling point.
#pragma once
#include "common.hpp"

namespace vllm {

// ---------------------------------------------------------------------------------------------------------
// Integer part of one sampling point -- shared by every kernel so that "index-exact" is a property of ONE
// function.  Mirrors ms_deform_im2col_cuda.cuh:277-292 and :38-41.
// ---------------------------------------------------------------------------------------------------------
// HIP's __fmul_rn / __fsub_rn are plain operators, and under the default -ffp-contract=fast the BACKEND fuses
// mul + sub into one fma whatever the source says (found on hardware: 1-2 ulp different h_im where loc*H crosses a
// power of two; `#pragma clang fp contract(off)` does not stop it).  The reference rounds twice and floor() of this
// value is the index-exact part of the contract, so the product is made opaque to the optimiser (zero instructions).
__device__ __forceinline__ float mul_rn(float a, float b)
{
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ double mul_rn(double a, double b)
{
    double p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ double sub_rn(double a, double b) { return a - b; }
__device__ __forceinline__ float floor_t(float a) { return floorf(a); }
__device__ __forceinline__ double floor_t(double a) { return floor(a); }

template <typename T>
struct SamplePoint {
    T h_im, w_im;
    int h_low, w_low;
    bool ok;
};

template <typename T>
__device__ __forceinline__ SamplePoint<T> sample_point(T loc_w, T loc_h, int H, int W)
{
    SamplePoint<T> s;
    s.h_im = sub_rn(mul_rn(loc_h, (T)H), (T)0.5);
    s.w_im = sub_rn(mul_rn(loc_w, (T)W), (T)0.5);
    s.ok = (s.h_im > (T)-1) && (s.w_im > (T)-1) && (s.h_im < (T)H) && (s.w_im < (T)W);
    // accepted coordinates lie in (-1, H) x (-1, W): their floor is in [-1, H-1].  A rejected coordinate may be NaN,
    // +-inf or huge; its float->int conversion (and the +1 that follows) must never reach address arithmetic.
    s.h_low = s.ok ? (int)floor_t(s.h_im) : 0;
    s.w_low = s.ok ? (int)floor_t(s.w_im) : 0;
    return s;
}

// Do the level maps form an exact 2x pyramid (H_l * 2^l == H_0, W_l * 2^l == W_0) whose cells are the Lq queries?
// Evaluated ON THE DEVICE from the shape tensor (block-uniform scalar loads), so that the launcher can enqueue the
// pyramid kernel (msda_tiled6.hip) and its general-geometry fallback back to back without a host synchronisation:
// exactly one of the two does the work.
__device__ __forceinline__ bool geometry_is_pyramid(const int64_t *shapes, int L, long Lq)
{
    if (L < 1 || L > 4) return false;
    const int H0 = (int)shapes[0], W0 = (int)shapes[1];
    // (the pyramid-item kernels pack (row + 1, column + 1) of a corner into 16-bit halves -- round 5 also reduces them as SIGNED 16-bit
    //  pairs: maps beyond 16384 pixels a side are "general geometry" and take generation 4)
    bool ok = H0 > 0 && W0 > 0 && H0 <= 16384 && W0 <= 16384;
    long cum = (long)H0 * W0;
    for (int l = 1; l < L; ++l) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        ok = ok && (Hl << l) == H0 && (Wl << l) == W0;
        cum += (long)Hl * Wl;
    }
    return ok && cum == Lq;
}

// The generation-7 kernel only needs NESTED maps: level l is the previous one halved, rounded either way (what strided
// convolutions / ceil-mode poolings give a detection backbone: 100 x 167 -> 50 x 84 -> 25 x 42 -> 13 x 21), every query of level
// l then belongs to exactly one 8 x 16 item of level 0 (row y >> (3 - l), column x >> (4 - l)) and every level-l query of an
// item exists or is masked.  Exact 2x pyramids are the special case without masked cells.
__device__ __forceinline__ bool geometry_is_nested(const int64_t *shapes, int L, long Lq)
{
    if (L < 1 || L > 4) return false;
    int Hp = (int)shapes[0], Wp = (int)shapes[1];
    bool ok = Hp > 0 && Wp > 0 && Hp <= 16384 && Wp <= 16384;   // (see geometry_is_pyramid)
    const int nty = (Hp + 7) >> 3, ntx = (Wp + 15) >> 4;
    long cum = (long)Hp * Wp;
    for (int l = 1; l < L; ++l) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        ok = ok && Hl > 0 && Wl > 0 && Hl >= (Hp >> 1) && Hl <= ((Hp + 1) >> 1) && Wl >= (Wp >> 1) && Wl <= ((Wp + 1) >> 1) &&
             Hl <= nty * (8 >> l) && Wl <= ntx * (16 >> l);
        cum += (long)Hl * Wl;
        Hp = Hl; Wp = Wl;
    }
    return ok && cum == Lq;
}

}  // namespace vllm
