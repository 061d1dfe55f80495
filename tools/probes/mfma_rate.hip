// Probe: issue rate of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 with NACC independent accumulators (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(float *out, int iters, long *cyc, const bf16x8_t *rnd)
{
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256, 1) void k16(float *out, int iters, long *cyc, const bf16x8_t *rnd)
{
    f32x4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// Same loops under __launch_bounds__(512, 1): the register budget is 256 per lane, so the accumulators stay in ArchVGPRs
// (under (256, 1) the compiler moves them to AccVGPRs: a[..] operands).  Launched with 256 or 512 threads.
template <int NACC>
__global__ __launch_bounds__(512, 1) void k32v(float *out, int iters, long *cyc, const bf16x8_t *rnd)
{
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512, 1) void k16v(float *out, int iters, long *cyc, const bf16x8_t *rnd)
{
    f32x4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename F> void run(const char *name, F kern, int nacc, double flops_per, const bf16x8_t *rnd = nullptr, int threads = 256)
{
    float *out; long *cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 10, cyc, rnd);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, cyc, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * nacc * (threads / 256);   // MFMAs per SIMD (the tick count is wave 0's own and only meaningful at one wave per SIMD)
    printf("%s%-28s acc=%2d  %.1f cyc/MFMA (s_memtime)  %.2f ns/MFMA  -> %.0f TF/s chip\n", rnd ? "[random bf16 data] " : "", name, nacc, c / n, ms * 1e6 / n,
           flops_per * n * 1024 / (ms * 1e-3) / 1e12);
}
int main()
{
    run("32x32x16 (16 regs/acc)", k32<1>, 1, 32768.0);
    run("32x32x16 (16 regs/acc)", k32<2>, 2, 32768.0);
    run("32x32x16 (16 regs/acc)", k32<3>, 3, 32768.0);
    run("32x32x16 (16 regs/acc)", k32<4>, 4, 32768.0);
    run("32x32x16 (16 regs/acc)", k32<8>, 8, 32768.0);
    run("32x32x16 (16 regs/acc)", k32<16>, 16, 32768.0);
    run("16x16x32 (4 regs/acc)", k16<1>, 1, 16384.0);
    run("16x16x32 (4 regs/acc)", k16<2>, 2, 16384.0);
    run("16x16x32 (4 regs/acc)", k16<4>, 4, 16384.0);
    run("16x16x32 (4 regs/acc)", k16<8>, 8, 16384.0);
    run("16x16x32 (4 regs/acc)", k16<16>, 16, 16384.0);
    run("16x16x32 (4 regs/acc)", k16<64>, 64, 16384.0);
    // random bf16 operands (normal-ish values): same instruction stream, realistic data toggling
    unsigned short h[512 * 8];
    unsigned x = 12345;
    for (int i = 0; i < 512 * 8; ++i) { x = x * 1664525u + 1013904223u; float f = ((int)(x >> 8) % 2001 - 1000) / 500.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    bf16x8_t *d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    run("32x32x16", k32<16>, 16, 32768.0, d);
    run("16x16x32", k16<64>, 64, 16384.0, d);
    printf("-- accumulators in ArchVGPRs (launch_bounds(512,1)), one wave per SIMD --\n");
    run("VGPR acc 16x16x32", k16v<2>, 2, 16384.0); run("VGPR acc 16x16x32", k16v<4>, 4, 16384.0); run("VGPR acc 16x16x32", k16v<8>, 8, 16384.0);
    run("VGPR acc 16x16x32", k16v<16>, 16, 16384.0); run("VGPR acc 16x16x32", k16v<32>, 32, 16384.0);
    run("VGPR acc 32x32x16", k32v<2>, 2, 32768.0); run("VGPR acc 32x32x16", k32v<8>, 8, 32768.0);
    run("VGPR acc 16x16x32", k16v<32>, 32, 16384.0, d); run("VGPR acc 32x32x16", k32v<8>, 8, 32768.0, d);
    printf("-- the same, TWO waves per SIMD (512-thread blocks; read the ns column) --\n");
    run("VGPR acc 16x16x32", k16v<16>, 16, 16384.0, nullptr, 512); run("VGPR acc 32x32x16", k32v<8>, 8, 32768.0, nullptr, 512);
    run("VGPR acc 16x16x32", k16v<16>, 16, 16384.0, d, 512); run("VGPR acc 32x32x16", k32v<8>, 8, 32768.0, d, 512);
    return 0;
}
