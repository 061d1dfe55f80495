// Hardware probes for gfx950 (run on the GPU box; prints to stdout).  Not part of the product.
//  1. MFMA 32x32x16 / 16x16x32 bf16 operand + accumulator layouts (asymmetric integer matrices vs CPU).
//  2. ds_read_b64_tr_b16 lane/element semantics (dump + check against the formula we will rely on).
//  3. global_load_lds (16 B) destination semantics.
//  4. vector-L1 / L2 / HBM gather bandwidth with the MSDA access shape (8 x 128 B lines per wave instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/probe.hip -o tools/probes/probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x)                                                                                              \
    do {                                                                                                   \
        hipError_t e = (x);                                                                                \
        if (e != hipSuccess) {                                                                             \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);                   \
            exit(1);                                                                                       \
        }                                                                                                  \
    } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)(u >> 16);
}

__global__ void mfma32_probe(const uint16_t *A /*[32][16]*/, const uint16_t *B /*[16][32]*/, float *C /*[32][32]*/)
{
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + e];    // A[i=l&31][k=8*(l>>5)+e]
        b[e] = (short)B[(8 * (l >> 5) + e) * 32 + (l & 31)];  // B[k][j=l&31]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        C[row * 32 + col] = c[r];
    }
}

__global__ void mfma16_probe(const uint16_t *A /*[16][32]*/, const uint16_t *B /*[32][16]*/, float *C /*[16][16]*/)
{
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + e];    // A[i=l&15][k=8*(l>>4)+e]
        b[e] = (short)B[(8 * (l >> 4) + e) * 16 + (l & 15)];  // B[k][j=l&15]
    }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// tr-read: LDS element value == element index; every lane supplies the address of 4 contiguous elements.
// mode 0: lane i of a 16-lane group points at row (i>>2), cols 4*(i&3) of a [4][row_stride] block.
__global__ void trread_probe(int *out /*[64][4]*/, int row_stride)
{
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int g = l >> 4, i = l & 15;
    const int elem = g * 1024 + (i >> 2) * row_stride + 4 * (i & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (int)(unsigned short)v[j];
}

// global_load_lds 16 B: does lane l land at lds_base + 16*l ?
__global__ void glds_probe(const uint32_t *src /*[64*4]*/, uint32_t *out /*[64*4]*/)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    // per-lane SOURCE address is permuted (lane l reads chunk 63-l); destination is wave-uniform base (+64 B).
    const uint32_t *gp = src + (63 - threadIdx.x) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                     (__attribute__((address_space(3))) void *)(lds + 16), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[16 + i];
}

// Gather bandwidth: each 8-lane group reads 128 contiguous bytes (16 B per lane) at a pseudo-random line of a
// `span`-byte window; ILP independent loads per iteration.
template <int ILP>
__global__ __launch_bounds__(256) void gather_bw(const float4 *base, size_t span_lines, int iters, float *sink,
                                                 unsigned seed)
{
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    unsigned s = seed + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 7919u + grp * 104729u;
    float acc = 0.f;
    // each block works inside its own window so that `span` is a per-CU (L1) footprint
    const float4 *win = base + (size_t)blockIdx.x * span_lines * 8;
    for (int it = 0; it < iters; ++it) {
        float4 v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            s = s * 1664525u + 1013904223u;
            const size_t line = (s >> 8) % span_lines;
            v[u] = win[line * 8 + sub];
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1.2345f) sink[0] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz  L2 %d  LDS/blk %zu\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.l2CacheSize, prop.sharedMemPerBlock);

    // ---- 1. MFMA layouts ----
    {
        std::vector<uint16_t> A(32 * 16), B(16 * 32);
        std::vector<float> Af(32 * 16), Bf(16 * 32), C(32 * 32), R(32 * 32, 0.f);
        srand(1);
        for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)(rand() % 9 - 4); A[i] = f2bf(Af[i]); }
        for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)(rand() % 7 - 3); B[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
        uint16_t *dA, *dB; float *dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        mfma32_probe<<<1, 64>>>(dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32 * 32; ++i) bad += (C[i] != R[i]);
        printf("MFMA32x32x16 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    }
    {
        std::vector<uint16_t> A(16 * 32), B(32 * 16);
        std::vector<float> Af(16 * 32), Bf(32 * 16), C(16 * 16), R(16 * 16, 0.f);
        srand(2);
        for (int i = 0; i < 16 * 32; ++i) { Af[i] = (float)(rand() % 9 - 4); A[i] = f2bf(Af[i]); }
        for (int i = 0; i < 32 * 16; ++i) { Bf[i] = (float)(rand() % 7 - 3); B[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
        uint16_t *dA, *dB; float *dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        mfma16_probe<<<1, 64>>>(dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 16 * 16; ++i) bad += (C[i] != R[i]);
        printf("MFMA16x16x32 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    }
    // ---- 2. tr read ----
    for (int stride : {16, 64, 72}) {
        int *d; std::vector<int> h(256);
        CK(hipMalloc(&d, 256 * 4));
        trread_probe<<<1, 64>>>(d, stride);
        CK(hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost));
        // expectation: lane (g,i) elem j == value at row j, col i of the block = g*1024 + j*stride + i
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += (h[l * 4 + j] != (l >> 4) * 1024 + j * stride + (l & 15));
        printf("tr_b16 (row stride %d) formula result[lane i][j] == block[row j][col i]: %s (%d mismatches)\n", stride,
               bad ? "FAIL" : "OK", bad);
        if (bad || stride == 16) {
            for (int l = 0; l < 20; ++l) printf("  lane %2d: %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        }
    }
    // ---- 3. global_load_lds ----
    {
        std::vector<uint32_t> src(256), out(256);
        for (int i = 0; i < 256; ++i) src[i] = i;
        uint32_t *ds, *dout;
        CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dout, 1024));
        CK(hipMemcpy(ds, src.data(), 1024, hipMemcpyHostToDevice));
        glds_probe<<<1, 64>>>(ds, dout);
        CK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int w = 0; w < 4; ++w) bad += (out[l * 4 + w] != (uint32_t)((63 - l) * 4 + w));
        printf("global_load_lds16: lds[base + 16*lane] <- src(lane): %s (%d mismatches) first words %u %u %u %u %u\n",
               bad ? "FAIL" : "OK", bad, out[0], out[1], out[4], out[8], out[252]);
    }
    // ---- 4. gather bandwidth ----
    {
        const size_t bytes = (size_t)1 << 31;  // 2 GiB arena
        float4 *buf; float *sink;
        CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
        CK(hipMemset(buf, 0, bytes));
        const int blocks = prop.multiProcessorCount * 4;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        // per-block window sizes: 8 KiB (L1 hit), 256 KiB (L2), 2 MiB (L2/MALL mix)
        const size_t spans[] = {8 << 10, 16 << 10, 64 << 10, 256 << 10, 2 << 20};
        for (size_t sp : spans) {
            const size_t lines = sp / 128;
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                gather_bw<8><<<blocks, 256>>>(buf, lines, iters, sink, 17u + rep);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double gb = (double)blocks * 256 * 16.0 * 8 * iters / 1e9;
                if (rep == 1)
                    printf("gather 128B-lines, per-block window %7zu B (4 blocks/CU): %.1f GB in %.3f ms = %.2f TB/s = %.1f B/clk/CU @2.4GHz\n",
                           sp, gb, ms, gb / ms, gb / ms * 1e12 / 1e3 / prop.multiProcessorCount / 2.4e9 * 1e0);
            }
        }
    }
    return 0;
}
