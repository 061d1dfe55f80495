// Shared host/device helpers for libvllm_hip.so (gfx950 only; no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/vllm_hip.h"

namespace vllm {

void set_error(const char *fmt, ...);

#define VLLM_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            ::vllm::set_error(__VA_ARGS__);      \
            return VLLM_EINVAL;                  \
        }                                        \
    } while (0)

// hipGetLastError() is sticky per thread: a benign failure inside the host framework (e.g. a capability probe)
// would otherwise be reported as OUR launch failing, so every launch goes through VLLM_LAUNCH (clear, then launch).
#define VLLM_LAUNCH(kernel, grid, block, lds, st, ...)                         \
    do {                                                                       \
        (void)hipGetLastError();                                               \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);         \
    } while (0)

#define VLLM_CHECK_LAUNCH(what)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            ::vllm::set_error("%s: %s", what, hipGetErrorString(e__));                 \
            return VLLM_ELAUNCH;                                                       \
        }                                                                              \
    } while (0)

// hipFuncSetAttribute (dynamic LDS above 64 KiB) is per DEVICE: a process that drives several GPUs must set it on each of them.
// Returns true the first time it is called on the current device for the given mask (thread-safe).
static inline bool first_use_on_device(unsigned long long *mask)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    const unsigned long long bit = 1ull << (d & 63);
    const unsigned long long old = __atomic_fetch_or(mask, bit, __ATOMIC_RELAXED);
    return !(old & bit);
}
// Compute units of the CURRENT device, cached per device (a `static int cus` filled from the first device queried is
// wrong for a second, smaller device; the persistent kernels size their grids with it).
static inline int device_cus()
{
    static int cache[64];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    int v = __atomic_load_n(&cache[d & 63], __ATOMIC_RELAXED);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        __atomic_store_n(&cache[d & 63], v, __ATOMIC_RELAXED);
    }
    return v;
}
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 (device) -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
// f32 -> bf16, round-to-nearest-even: let the compiler select v_cvt_pk_bf16_f32 (gfx950) -- it also inserts the
// wait state the trans-op -> VALU hazard needs (a hand-written asm cvt right behind v_exp_f32 read a stale value).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    const float2v_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

}  // namespace vllm
