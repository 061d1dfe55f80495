// Shared by msda.hip and msda_tiled.hip: the ONE definition of the integer part of a sampling point.
#pragma once
#include "common.hpp"

namespace vllm {

// ---------------------------------------------------------------------------------------------------------
// Integer part of one sampling point -- shared by every kernel so that "index-exact" is a property of ONE
// function.  Mirrors ms_deform_im2col_cuda.cuh:277-292 and :38-41.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
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
    // floor of a rejected (possibly NaN / huge) coordinate is never used for addressing un-clamped.
    s.h_low = (int)floor_t(s.h_im);
    s.w_low = (int)floor_t(s.w_im);
    return s;
}

}  // namespace vllm
