// Probe (round 3): VALU issue rate per SIMD as a function of the number of waves on the SIMD.
//   Every CU runs ONE block of 256 * WPS threads (WPS waves per SIMD; 100 KB of dynamic LDS keeps a second block away);
//   a wave executes ITERS x 64 instructions of one opcode over 8 independent register chains (dependency distance 8
//   instructions), the loop overhead (s_add / s_cmp / s_cbranch per 64 VALU) is < 5 %.
//   Reported: shader cycles (s_memtime) per instruction PER SIMD = cycles * 1 / (instructions per wave * WPS), and the
//   same from the event-timed wall clock at the clock rocm-smi reports under the load (printed as ns).
// Question it settles: is a wave64 VALU instruction 2 cycles (SIMD-32, the guide) or 4 (what a ONE-wave probe sees)?
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

enum { OP_FMA, OP_PKFMA, OP_ADDU, OP_FMAC, OP_EXP, OP_MOVDPP, OP_FMA_SGPR, NOPS };
static const char *OPN[NOPS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_fmac_f32", "v_exp_f32", "v_mov_b32 dpp quad_perm", "v_fma_f32 (sgpr operand)"};

template <int OP> __device__ __forceinline__ void step(float (&x)[8], float2 (&x2)[8], float c, float2 c2, float sc)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
        if (OP == OP_FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "s"(sc));
        if (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x[i]) : "v"(c));
        if (OP == OP_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
        if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (OP == OP_MOVDPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
        if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x2[i]) : "v"(c2));
    }
}

template <int OP, int WPS> __global__ __launch_bounds__(256 * WPS) void k_issue(float *out, int iters, long long *cyc)
{
    extern __shared__ char hog[];
    float x[8];
    float2 x2[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; x2[i] = make_float2(x[i], x[i] + 1.f); }
    const float c = 1.0001f;
    const float2 c2 = make_float2(c, c);
    const float sc = __builtin_amdgcn_readfirstlane(iters) * 1e-9f + 1.0f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) step<OP>(x, x2, c, c2, sc);   // 64 instructions
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + x2[i].x + x2[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + (hog[0] ? 1.f : 0.f);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float *g_out;
static long long *g_cyc;
template <int OP, int WPS> static void run()
{
    const int iters = 20000;
    auto kern = k_issue<OP, WPS>;
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best_ms = 1e9;
    long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WPS), lds, 0, g_out, iters, g_cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_ms) { best_ms = ms; hipMemcpy(&cyc, g_cyc, 8, hipMemcpyDeviceToHost); }
    }
    const double n = (double)iters * 64.0;           // instructions per wave
    const double per_simd = n * WPS;                  // instructions issued on one SIMD
    printf("%-26s waves/SIMD %d: %6.2f memtime ticks / inst / SIMD   %6.3f ns / inst / SIMD   (%.2f ticks per wave-inst)\n", OPN[OP], WPS,
           cyc / per_simd, best_ms * 1e6 / per_simd, cyc / n);
}
template <int OP> static void sweep() { run<OP, 1>(); run<OP, 2>(); run<OP, 3>(); run<OP, 4>(); }

int main()
{
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_cyc, 64);
    printf("# s_memtime ticks: 100 MHz 'memrealtime' would read 0.04x of these; compare with the ns column (2.4 GHz: 1 cycle = 0.417 ns)\n");
    sweep<OP_FMA>(); sweep<OP_FMA_SGPR>(); sweep<OP_FMAC>(); sweep<OP_PKFMA>(); sweep<OP_ADDU>(); sweep<OP_MOVDPP>(); sweep<OP_EXP>();
    return 0;
}
