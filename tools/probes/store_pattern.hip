// Probe: cost of the GEMM epilogue's store pattern.  256 blocks x 512 threads each write a 256x256 bf16 tile of a
// [4096, 4096] matrix: (A) as gemm256 does (a wave instruction = 16 rows x 32 contiguous bytes, 8 B per lane),
// (B) row-contiguous (a wave instruction = 2 rows x 512 contiguous bytes, 16 B per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void store_a(uint16_t *y, int ld)
{
    const int tile = blockIdx.x, tm = tile / 16, tn = tile % 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wr = wave >> 2, wc = wave & 3, fr = lane & 15, kq = lane >> 4;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 4; ++j) {
                const int n = tn * 256 + (q & 1) * 128 + wc * 32 + i * 16 + kq * 4;
                const int m = tm * 256 + (q >> 1) * 128 + wr * 64 + j * 16 + fr;
                u32x2 v = {(uint32_t)(m * 3 + n), (uint32_t)(m + n * 5)};
                *reinterpret_cast<u32x2 *>(y + (size_t)m * ld + n) = v;
            }
}
__global__ __launch_bounds__(512) void store_b(uint16_t *y, int ld)
{
    const int tile = blockIdx.x, tm = tile / 16, tn = tile % 16;
    const int t = threadIdx.x;   // 32 lanes x 16 B = one 512-byte tile row; 16 rows per pass, 16 passes
    for (int p = 0; p < 16; ++p) {
        const int m = tm * 256 + p * 16 + (t >> 5), n = tn * 256 + (t & 31) * 8;
        u32x4 v = {(uint32_t)(m * 3 + n), (uint32_t)(m + n * 5), (uint32_t)m, (uint32_t)n};
        *reinterpret_cast<u32x4 *>(y + (size_t)m * ld + n) = v;
    }
}
template <typename K> float timeit(K k, uint16_t *y)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, y, 4096);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, y, 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}
int main()
{
    uint16_t *y; hipMalloc(&y, 4096UL * 4096 * 2);
    printf("pattern A (16 rows x 32 B per wave store, 8 B/lane):  %.1f us per 32 MB\n", timeit(store_a, y));
    printf("pattern B (2 rows x 512 B per wave store, 16 B/lane): %.1f us per 32 MB\n", timeit(store_b, y));
    return 0;
}
