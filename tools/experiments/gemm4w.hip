// bf16 GEMM, 256x256x32 block tile, FOUR waves, one wave per SIMD with a 128x128 register tile ("gemm_variant" 3).
//
// Same contract and epilogues as gemm.hip / gemm256.hip (Y = epi(X W^T + b), nn.Linear layout).  Why a second schedule:
// PMC runs of gemm256 next to hipBLASLt on the same shapes (tools/pmc_gemm_run.sh) show equal MFMA work but 1.7x the LDS
// instructions and 2.2x the wait cycles in the 8-wave kernel (MFMA pipe 72 % vs 85 % busy inside the kernel).  A 128x128
// wave tile reads (128 + 128) x k per step instead of (128 + 64) x k for half the MFMAs, and four waves meet at ONE
// barrier per K tile instead of eight.
//   * wave (wr, wc) owns outputs m in [wr*128, +128), n in [wc*128, +128): 4 x 4 blocks of v_mfma_f32_32x32x16_bf16,
//     256 fp32 accumulators per lane (the compiler places them in AGPRs; 512 registers at one wave per SIMD);
//   * swapped product (A operand = W rows, B operand = X rows): a lane ends up with 4 consecutive output features of one
//     token per accumulator quad -> the shared epilogue (gemm_epilogue.hpp) stores 8 bytes at a time;
//   * LDS: 4 stages x {X tile, W tile} of 256 rows x 32 k (16 KiB each) = 128 KiB, filled by LDS-DMA, 8 instructions per
//     wave per K tile.  Rows are 64 B; the 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3) (source-side
//     swizzle), which makes every ds_read_b128 lane group (16 rows x 1 chunk) hit 64 distinct banks;
//   * per K tile (two k16 steps): fragments of step 1 are read while step 0 multiplies; in the middle of the tile the
//     wave retires its reads (lgkmcnt 0) and its share of tile t+1 (COUNTED vmcnt(8): tile t+2 stays in flight), meets
//     the block at the tile's only barrier, refills the stage of tile t-1 with tile t+3 and reads step 0 of tile t+1
//     while step 1 multiplies.  Three K tiles are in flight; loads have two tile times (~2000 cycles) to land.
// STATUS (round 1): correct (same tests as the other schedules) but NOT the default -- 970-1010 TFLOP/s at 4096^3 and
// 630-710 on the K=1024 ViT-L shapes against 1290 / 760-950 for gemm256.  Ablations at 4096^3 (160 us): no LDS-DMA -30 us,
// no fragment reads -8, no barrier -4.5, bare MFMA stream + prologue/epilogue 123.5 of which the MFMAs are 75: (1) an
// LDS-DMA instruction costs the issuing wave 60-185 cycles (guide, MI355X_MICROARCH.md) and with one wave per SIMD
// nothing covers it -> the refill has to go global -> VGPR -> ds_write; (2) the epilogue (32 rows x 16 B per store
// instruction, nothing to overlap it with) costs ~36 us per round of tiles against ~16 us in gemm256 -> it has to go
// through LDS and leave as 16-byte row-contiguous stores.  Both are round-2 work; the schedule itself keeps the MFMA
// stream clean (256 accumulators in AGPRs, no moves in the loop).
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int W4_BM = 256, W4_BN = 256, W4_BK = 32;
constexpr int W4_THREADS = 256;
constexpr int W4_TILE = 256 * W4_BK * 2;   // 16 KiB: 256 rows x 64 B
constexpr int W4_STAGE = 2 * W4_TILE;      // X tile | W tile
constexpr int W4_STAGES = 4;

template <int EPI>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm4w_bf16_kernel(const GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 stages x 32 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- XCD-aware tile mapping (as gemm.hip / gemm256.hip) ----
    int tm_idx, tn_idx;
    {
        const int tile = blockIdx.x, xcd = tile & 7, s = tile >> 3;
        if ((a.nt & 7) == 0) {
            const int npx = a.nt >> 3;
            tn_idx = xcd + 8 * (s % npx);
            tm_idx = s / npx;
        } else {
            tm_idx = xcd + 8 * (s / a.nt);
            tn_idx = s % a.nt;
        }
        if (tm_idx >= a.mt || tn_idx >= a.nt) return;
    }
    const int m0 = tm_idx * W4_BM, n0 = tn_idx * W4_BN;
    const int nk = a.K / W4_BK;

    f32x16_t acc[4][4];   // [n block i][m block j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- loader: 16 segments of 16 rows per tile half; wave w moves segments 4w .. 4w+3 of both halves ----
    const uint16_t *xsrc[4], *wsrc[4];
    {
        const int rl = lane >> 2, pos = lane & 3;
        const int c = pos ^ ((rl >> 2) & 3);            // global chunk this lane fetches (its LDS position is `pos`)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = (wave * 4 + s) * 16 + rl;
            int gm = m0 + row;
            gm = gm < a.M ? gm : a.M - 1;
            if (a.xP > 0) gm += gm / a.xP + 1;
            int gn = n0 + row;
            gn = gn < a.N ? gn : a.N - 1;
            xsrc[s] = a.X + (size_t)gm * a.ldx + c * 8;
            wsrc[s] = a.W + (size_t)gn * a.ldw + c * 8;
        }
    }
    auto issue = [&](int t) {   // K tile t -> stage t & 3 (clamped: refills past the end are harmless and keep the counts)
        const int k0 = (t < nk ? t : nk - 1) * W4_BK;
        char *st = smem + (t & (W4_STAGES - 1)) * W4_STAGE + wave * 4096;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[s] + k0),
                                             (__attribute__((address_space(3))) void *)(st + s * 1024), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[s] + k0),
                                             (__attribute__((address_space(3))) void *)(st + W4_TILE + s * 1024), 16, 0, 0);
    };

    // ---- fragment addresses (byte offsets inside a stage); k16 step 1 is the same address with chunk bit 1 flipped ----
    int xoff[4], woff[4];
    {
        const int g = (l31 >> 2) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) xoff[j] = (wr * 128 + j * 32 + l31) * 64 + ((hi ^ g) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) woff[i] = W4_TILE + (wc * 128 + i * 32 + l31) * 64 + ((hi ^ g) << 4);
    }
    bf16x8_t xf[2][4], wf[2][4];
    auto read_frags = [&](int t, int ks) {
        const char *st = smem + (t & (W4_STAGES - 1)) * W4_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8_t *>(st + (woff[i] ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[ks][j] = *reinterpret_cast<const bf16x8_t *>(st + (xoff[j] ^ (ks << 5)));
    };
#define W4_MMA_ROWS(KS, I0, I1)                                                                             \
    do {                                                                                                    \
        _Pragma("unroll") for (int i = I0; i < I1; ++i)                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[KS][i], xf[KS][j], acc[i][j], 0, 0, 0); \
    } while (0)

    // ---- prologue: tiles 0, 1, 2 in flight; tile 0 landed for everybody; step 0 of tile 0 in registers ----
    issue(0); issue(1); issue(2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0);

    // One wave per SIMD: nothing else can feed the matrix pipe while this wave issues memory instructions, so they are
    // spread between the MFMAs (sched_group_barrier: MFMA 0x8, VMEM read 0x20, DS read 0x100).  First half of a tile:
    // the 8 fragment reads of step 1 ride on the MFMAs of step 0; second half (behind the tile's barrier): the 8 LDS-DMA
    // refills, then the 8 fragment reads of the next tile's step 0, one per MFMA of step 1.
    for (int t = 0; t < nk; ++t) {
        read_frags(t, 1);
        W4_MMA_ROWS(0, 0, 4);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of tile t are done (its stage may be refilled)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // my share of tile t+1 has landed (tile t+2 stays in flight)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(t + 3);                                        // into the stage of tile t-1
        read_frags(t + 1, 0);
        W4_MMA_ROWS(1, 0, 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#undef W4_MMA_ROWS

    // ---- epilogue: accumulator quad q of block (i, j) = features n .. n+3 of token m ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wc * 128 + i * 32 + 8 * q + 4 * hi;
            if (n >= a.N) continue;
            const EpiCols cols = epi_cols<EPI>(a, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wr * 128 + j * 32 + l31;
                if (m >= a.M) continue;
                const float v4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                epi_store<EPI>(a, m, n, v4, cols);
            }
        }
}

int gemm4w_bf16_launch(int epi, GemmArgs a, hipStream_t st)
{
    a.mt = ceil_div(a.M, W4_BM);
    a.nt = ceil_div(a.N, W4_BN);
    long tiles;
    if ((a.nt & 7) == 0) tiles = (long)a.mt * a.nt;
    else tiles = (long)((a.mt + 7) / 8) * 8 * a.nt;
    const dim3 grid((unsigned)tiles), block(W4_THREADS);
    const size_t lds = W4_STAGES * W4_STAGE;   // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
#define SETATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm4w_bf16_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        SETATTR(EPI_BIAS); SETATTR(EPI_GELU); SETATTR(EPI_QUICK_GELU); SETATTR(EPI_RESIDUAL); SETATTR(EPI_EMBED); SETATTR(EPI_F32);
#undef SETATTR
        attr_set = true;
    }
#define L(E) VLLM_LAUNCH((gemm4w_bf16_kernel<E>), grid, block, lds, st, a)
    switch (epi) {
    case EPI_BIAS: L(EPI_BIAS); break;
    case EPI_GELU: L(EPI_GELU); break;
    case EPI_QUICK_GELU: L(EPI_QUICK_GELU); break;
    case EPI_RESIDUAL: L(EPI_RESIDUAL); break;
    case EPI_EMBED: L(EPI_EMBED); break;
    case EPI_F32: L(EPI_F32); break;
    default: set_error("gemm4w: unknown epilogue %d", epi); return VLLM_EINVAL;
    }
#undef L
    VLLM_CHECK_LAUNCH("gemm4w_bf16_kernel");
    return VLLM_OK;
}

}  // namespace vllm
