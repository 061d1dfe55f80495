// Probe: do MFMAs of one wave overlap VALU work of ANOTHER wave on the same SIMD?  Block = 8 waves (2 per SIMD):
// waves 0-3 issue NM MFMA 32x32x16 per iteration, waves 4-7 issue NV unrolled VALU ops (8 independent chains).
// Reports ns per iteration for A alone, B alone, both; "sum" vs "max" tells exclusive vs overlapped issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int NM, int NV, int KIND> __global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    const float c = 1.0001f;
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NM; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(c));
                    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
                    if (KIND == 2) { if (j % 3 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7])); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(c)); }
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
static float *g_out;
template <int NM, int NV, int KIND> static double t(int mode)
{
    const int iters = 1000;
    for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(256), dim3(512), 0, 0, g_out, iters, mode); }
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(256), dim3(512), 0, 0, g_out, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}
template <int NM, int NV, int KIND> static void row(const char *kind)
{
    const double a = t<NM, NV, KIND>(1), b = t<NM, NV, KIND>(2), ab = t<NM, NV, KIND>(3);
    printf("A %2d MFMA32 | B %3d %-8s : A %.0f ns  B %.0f ns  both %.0f ns   (sum %.0f, max %.0f)\n", NM, NV, kind, a, b, ab, a + b, a > b ? a : b);
}
int main()
{
    (void)hipMalloc(&g_out, 256 * 512 * 4);
    row<16, 64, 0>("fma"); row<16, 128, 0>("fma"); row<16, 192, 0>("fma"); row<16, 256, 0>("fma");
    row<16, 32, 1>("exp"); row<16, 64, 1>("exp"); row<16, 128, 1>("exp");
    row<16, 96, 2>("fma+exp"); row<16, 192, 2>("fma+exp");
    return 0;
}
