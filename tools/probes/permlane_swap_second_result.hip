// Compile-only probe (round 3): what this toolchain makes of the TWO results of __builtin_amdgcn_permlane16_swap when they are used
// separately.  v_permlane16_swap_b32 vdst, src0 exchanges the odd lane rows of vdst with the even lane rows of src0 and writes
// BOTH registers; the builtin returns {new vdst, new src0}.
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only permlane_swap_second_result.hip -o -
// Seen (ROCm 7.2.0): the instruction is emitted as `v_permlane16_swap_b32 v1, v2`, then r.x + 5 is computed from v1 INTO v2 and
// r.y * 7 from v1 again -- the second result (v2) is never read and is overwritten.  gemm256p.hip does the exchange in inline
// assembly with both registers as read-write operands.
#include <hip/hip_runtime.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *p, float *q)
{
    float a = p[threadIdx.x] + 1.0f, b = p[threadIdx.x + 64] * 3.0f;
    u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    q[threadIdx.x] = __builtin_bit_cast(float, r.x) + 5.0f;
    q[threadIdx.x + 64] = __builtin_bit_cast(float, r.y) * 7.0f;
}
