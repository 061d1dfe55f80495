// Probe: VALU / transcendental issue cost on gfx950 and how much of it hides under MFMAs
//   (a) one wave per SIMD, NCH independent chains of one opcode: cycles per instruction;
//   (b) one wave per SIMD, 1 MFMA 32x32x16 followed by K independent VALU ops: cycles per group (what fits "under" an MFMA);
//   (c) two waves per SIMD, one MFMA-only, one VALU-only: does the pair overlap.
// Numbers quoted in NOTES/rounds_1_to_4.md section 3.3 (attention "what bounds it").
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

enum { OP_EXP, OP_FMA, OP_MAX3, OP_CVT, OP_PKMUL, OP_ADD, OP_SWAP32, OP_BPERM, OP_PKFMA, OP_LOG, OP_RCP, NOPS };
static const char *OPN[NOPS] = {"v_exp_f32", "v_fma_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_add_f32",
                                "v_permlane32_swap", "ds_bpermute_b32", "v_pk_fma_f32", "v_log_f32", "v_rcp_f32"};

template <int OP> __device__ __forceinline__ void one(float &x, float &y, float c)
{
    if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(x));
    if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(y));
    if (OP == OP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == OP_SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    if (OP == OP_BPERM) { asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(y)); }
}
template <int OP> __device__ __forceinline__ void one2(float2 &x, float2 c)
{
    if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
}

template <int OP, int NCH> __global__ __launch_bounds__(256, 1) void k_rate(float *out, int iters, long *cyc)
{
    float x[NCH], y[NCH];
    float2 x2[NCH];
    for (int i = 0; i < NCH; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = (threadIdx.x * 4) & 255; x2[i] = make_float2(x[i], y[i]); }
    const float c = 1.0001f;
    const float2 c2 = make_float2(c, c);
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (OP == OP_PKMUL || OP == OP_PKFMA) one2<OP>(x2[i], c2); else one<OP>(x[i], y[i], c);
        }
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += x[i] + y[i] + x2[i].x + x2[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// (b) 1 MFMA + K VALU ops (OP) in the same wave, all independent
template <int OP, int K> __global__ __launch_bounds__(256, 1) void k_under(float *out, int iters, long *cyc)
{
    f32x16_t acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = i; }
    const float c = 1.0001f;
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) one<OP>(x[i & 7], y[i & 7], c);
        }
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// (c) 8 waves per CU: waves 0-3 MFMA only (nm per iteration), waves 4-7 VALU only (nv of OP per iteration)
template <int OP> __global__ __launch_bounds__(512, 2) void k_pair(float *out, int iters, int nm, int nv, long *cyc)
{
    const int wave = threadIdx.x >> 6;
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)(threadIdx.x * 3)};
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = i; }
    const float c = 1.0001f;
    const long t0 = clock64();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it)
            for (int j = 0; j < nm; j += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
            }
    } else {
        for (int it = 0; it < iters; ++it)
            for (int j = 0; j < nv; j += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) one<OP>(x[i], y[i], c);
            }
    }
    const long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

static float *g_out; static long *g_cyc;
template <typename F, typename... A> static double launch(F kern, int threads, A... args)
{
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, args...);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6;   // ns
}
template <int OP, int NCH> static void rate()
{
    const int iters = 4000;
    const double ns = launch(k_rate<OP, NCH>, 256, g_out, iters, g_cyc);
    long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NCH;
    printf("(a) %-20s chains=%2d  %.2f memtime-ticks/inst  %.3f ns/inst\n", OPN[OP], NCH, c / n, ns / n);
}
template <int OP, int K> static void under()
{
    const int iters = 4000;
    const double ns = launch(k_under<OP, K>, 256, g_out, iters, g_cyc);
    const double n = (double)iters * 2;
    printf("(b) MFMA32 + %2d x %-18s %.2f ns/group\n", K, OPN[OP], ns / n);
}
template <int OP> static void pair(int nm, int nv)
{
    const int iters = 2000;
    const double ns = launch(k_pair<OP>, 512, g_out, iters, nm, nv, g_cyc);
    printf("(c) waveA %3d MFMA32 | waveB %3d x %-18s %.2f ns/iter\n", nm, nv, OPN[OP], ns / iters);
}
int main()
{
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 64);
    rate<OP_FMA, 8>(); rate<OP_FMA, 1>(); rate<OP_ADD, 8>(); rate<OP_EXP, 8>(); rate<OP_EXP, 1>(); rate<OP_LOG, 8>(); rate<OP_RCP, 8>();
    rate<OP_MAX3, 8>(); rate<OP_MAX3, 1>(); rate<OP_CVT, 8>(); rate<OP_PKMUL, 8>(); rate<OP_PKFMA, 8>(); rate<OP_SWAP32, 8>(); rate<OP_BPERM, 8>();
    rate<OP_BPERM, 1>();
    under<OP_FMA, 0>(); under<OP_FMA, 4>(); under<OP_FMA, 6>(); under<OP_FMA, 7>(); under<OP_FMA, 8>(); under<OP_FMA, 12>();
    under<OP_EXP, 1>(); under<OP_EXP, 2>(); under<OP_EXP, 3>(); under<OP_EXP, 4>(); under<OP_EXP, 6>(); under<OP_EXP, 8>();
    under<OP_CVT, 6>(); under<OP_MAX3, 6>();
    pair<OP_FMA>(16, 0); pair<OP_FMA>(0, 128); pair<OP_FMA>(16, 128); pair<OP_FMA>(16, 64); pair<OP_FMA>(16, 96);
    pair<OP_EXP>(0, 32); pair<OP_EXP>(16, 32); pair<OP_EXP>(16, 64); pair<OP_EXP>(0, 64);
    return 0;
}
