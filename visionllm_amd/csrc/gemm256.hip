// bf16 GEMM, 256x256x64 block tile, 8 waves, 8-phase software pipeline -- the large-N linears (QKV, fc1).
//
// Same contract and epilogues as gemm.hip (Y = epi(X W^T + b)); this file is the high-throughput schedule:
//   * 8 waves as 2(M) x 4(N); a wave owns 128x64 outputs = four 64x32 pieces, one in each 128x128 block quadrant
//     (A-half i x B-half j), 32 v_mfma_f32_16x16x32_bf16 accumulators;
//   * LDS 128 KiB = 2 stages x {A0, A1, B0, B1} half-tiles of 128 rows x 64 k (16 KiB each), filled by LDS-DMA
//     (global_load_lds_dwordx4, 2 per wave per half-tile) with the source-side XOR swizzle of gemm.hip;
//   * every K tile is 4 phases, one block quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0).  After phase 1/2/3/4 the
//     half-tile A0/B1/A1/B0 has been read for the last time and is refilled ONE phase later with the data of two
//     K tiles ahead (B0: one tile ahead) -> three half-tiles are always in flight and the only vmcnt wait is a
//     COUNTED s_waitcnt vmcnt(6) once per K tile (never 0 in the main loop);
//   * the two wave groups (wr = 0 / 1; one wave of each per SIMD) run staggered by one barrier: while one group is
//     in its MFMA section (s_setprio 1) the other issues its ds_reads and LDS-DMA -> the matrix pipe of every SIMD
//     always has a wave feeding it.  lgkmcnt(0) sits BEFORE the phase's first barrier so a half-tile is never
//     refilled while a lagging wave still reads it (WAR), and reads of a refilled half-tile start one phase after
//     the vmcnt wait + barrier that retire it (RAW) -- /opt/skills/guides/cdna_hip_programming.md section 5.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int G2_BN = 256, G2_BK = 64;   // block rows are 64 * MT (template parameter)
constexpr int G2_THREADS = 512;
constexpr int G2_HALF = 128 * G2_BK * 2;          // 16 KiB half-tile
constexpr int G2_STAGE = 4 * G2_HALF;             // A0 A1 B0 B1
constexpr int OFF_A0 = 0, OFF_A1 = G2_HALF, OFF_B0 = 2 * G2_HALF, OFF_B1 = 3 * G2_HALF;

// Refill one half-tile of NSEG x 8 rows (NSEG <= 16): one LDS-DMA instruction per 8 rows, EXACTLY 2 per wave (the counted
// vmcnt waits assume that): waves whose segments do not exist (NSEG < 16) reload the last real segment into the unused
// tail of the 16 KiB slot.
template <int NSEG>
__device__ __forceinline__ void issue_half(const uint16_t *__restrict__ src, int ld, int row0, int nrows, int k0,
                                           char *lds_half, int wave, int lane, int skipP)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int seg = wave * 2 + s;
        const int sseg = seg < NSEG ? seg : NSEG - 1;      // source segment (dummy reload for idle slots)
        const int r = sseg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);
        int grow = row0 + r;
        grow = grow < nrows ? grow : nrows - 1;
        if (skipP > 0) grow += grow / skipP + 1;
        const uint16_t *g = src + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(lds_half + seg * 1024), 16, 0, 0);
    }
}

#define G2_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G2_BARRIER()                      \
    do {                                  \
        __builtin_amdgcn_s_barrier();     \
        __builtin_amdgcn_sched_barrier(0);\
    } while (0)

// MT = 16-row m tiles per wave per A half: 4 -> 256-row block tile, 3 -> 192 rows (better tile-count quantisation on
// 256 CUs for some shapes: 23080 x 1024 is 364 tiles = 1.42 rounds at 256 rows but 484 = 1.89 rounds at 192).
template <int EPI, int MT>
__global__ __launch_bounds__(G2_THREADS, 1) void gemm256_bf16_kernel(const GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 64 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, kq = lane >> 4;

    // ---- XCD-aware tile mapping (as gemm.hip) ----
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
    constexpr int BM_ = 64 * MT;                  // block rows: 2 halves x 2 wave rows x MT x 16
    const int m0 = tm_idx * BM_, n0 = tn_idx * G2_BN;
    const int nk = a.K / G2_BK;

    f32x4_t acc[4][2][MT];   // [quadrant q = 2*i + j][n tile][m tile]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[q][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS byte offsets inside a half-tile for the two fragment kinds (ks = 0 / 1 differ by XOR 4 chunks)
    int xoff[MT], woff[2];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = wr * (16 * MT) + t * 16 + fr;
        xoff[t] = r * 128 + ((kq ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = wc * 32 + t * 16 + fr;
        woff[t] = r * 128 + ((kq ^ (r & 7)) << 4);
    }
    bf16x8_t xf[MT][2], wf[2][2];

    auto k_of = [&](int t) { return (t < nk ? t : nk - 1) * G2_BK; };   // clamped: tail refills are harmless
    auto issue_A = [&](int half, int stage, int t) {
        issue_half<4 * MT>(a.X, a.ldx, m0 + half * (32 * MT), a.M, k_of(t), smem + stage * G2_STAGE + (half ? OFF_A1 : OFF_A0),
                           wave, lane, a.xP);
    };
    auto issue_B = [&](int half, int stage, int t) {
        issue_half<16>(a.W, a.ldw, n0 + half * 128, a.N, k_of(t), smem + stage * G2_STAGE + (half ? OFF_B1 : OFF_B0), wave,
                       lane, 0);
    };
    auto read_x = [&](const char *half) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            xf[t][0] = *reinterpret_cast<const bf16x8_t *>(half + xoff[t]);
            xf[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (xoff[t] ^ 64));   // chunk index + 4  (ks = 1)
        }
    };
    auto read_w = [&](const char *half) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            wf[t][0] = *reinterpret_cast<const bf16x8_t *>(half + woff[t]);
            wf[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (woff[t] ^ 64));
        }
    };
#define G2_MMA(Q)                                                                                               \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                  \
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], xf[j][ks], acc[Q][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    // ---- prologue: tile 0 complete in stage 0; tile 1's A0, B1, A1 in flight in stage 1 ----
    issue_A(0, 0, 0); issue_B(0, 0, 0); issue_B(1, 0, 0); issue_A(1, 0, 0);
    issue_A(0, 1, 1); issue_B(1, 1, 1); issue_A(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G2_BARRIER();
    if (wr == 1) G2_BARRIER();   // stagger: group 1 runs one barrier behind group 0

    for (int t = 0; t < nk; ++t) {
        const int s = t & 1;
        const char *st = smem + s * G2_STAGE;
        // phase 1: quadrant (A0,B0); refill B0 of the OTHER stage with tile t+1
        read_x(st + OFF_A0); read_w(st + OFF_B0);
        issue_B(0, s ^ 1, t + 1);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(0);
        G2_BARRIER();
        // phase 2: quadrant (A0,B1); refill A0 (this stage) with tile t+2
        read_w(st + OFF_B1);
        issue_A(0, s, t + 2);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(1);
        G2_BARRIER();
        // phase 3: quadrant (A1,B1); refill B1 with tile t+2
        read_x(st + OFF_A1);
        issue_B(1, s, t + 2);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(3);
        G2_BARRIER();
        // phase 4: quadrant (A1,B0); refill A1 with tile t+2; retire everything but the last 3 half-tiles
        read_w(st + OFF_B0);
        issue_A(1, s, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(2);
        G2_BARRIER();
    }
    if (wr == 0) G2_BARRIER();   // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup

    // ---- epilogue ----
    // bf16 outputs leave through LDS: the accumulator layout gives a wave store of 16 rows x 32 contiguous bytes
    // (11.1 us per 32 MB round of tiles, tools/probes/store_pattern.hip); re-read row-wise the tile leaves as 16-byte
    // lane stores, 512 contiguous bytes per row (7.1 us).  The pipeline's LDS is free by now; rows are padded to 528 B.
    constexpr int OPITCH = 528;
    const bool via_lds = !a.direct_store && EPI != EPI_F32 && (a.N & 7) == 0 && (a.ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15u) == 0;
    if (via_lds) {
        __builtin_amdgcn_s_barrier();   // every wave is out of the main loop: no fragment read of the ring is pending
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qi = q >> 1, qj = q & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = qj * 128 + wc * 32 + i * 16 + kq * 4;
                const int n = n0 + nl;
                const bool nok = n < a.N;
                const EpiCols cols = epi_cols<EPI>(a, nok ? n : 0);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int ml = qi * (32 * MT) + wr * (16 * MT) + j * 16 + fr;
                    const int m = m0 + ml;
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (nok && m < a.M) epi_value<EPI>(a, m, n, acc[q][i][j], cols, v);
                    uint2_t o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2_t *>(smem + ml * OPITCH + nl * 2) = o;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        const int t = threadIdx.x, c8 = (t & 31) * 8;   // 32 lanes x 16 B = one 512-byte tile row
#pragma unroll 4
        for (int p = 0; p < BM_ / 16; ++p) {
            const int ml = p * 16 + (t >> 5), m = m0 + ml, n = n0 + c8;
            if (m < a.M && n < a.N) {
                const uint4_t o = *reinterpret_cast<const uint4_t *>(smem + ml * OPITCH + c8 * 2);
                *reinterpret_cast<uint4_t *>(a.Y + epi_out_row<EPI>(a, m) * a.ldy + n) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int qi = q >> 1, qj = q & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + qj * 128 + wc * 32 + i * 16 + kq * 4;
            if (n >= a.N) continue;
            const EpiCols cols = epi_cols<EPI>(a, n);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int m = m0 + qi * (32 * MT) + wr * (16 * MT) + j * 16 + fr;
                if (m >= a.M) continue;
                epi_store<EPI>(a, m, n, acc[q][i][j], cols);
            }
        }
    }
}

int gemm256_bf16_launch(int epi, GemmArgs a, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    a.nt = ceil_div(a.N, G2_BN);
    // block rows 256 (MT=4) or 192 (MT=3): pick the one with the smaller (rounds x tile cost) on this many CUs
    // measured: a 192-row tile costs 0.87 of a 256-row tile (12 instead of 16 MFMAs per phase, same barriers)
    auto rounds_cost = [&](long rows, int mt_rows) {
        const long tiles = (long)ceil_div(rows, 64 * mt_rows) * a.nt;
        return ((tiles + cus - 1) / cus) * (long)(mt_rows == 4 ? 100 : 87);
    };
    int MT = rounds_cost(a.M, 3) < rounds_cost(a.M, 4) ? 3 : 4;
    if (a.variant256 == 3 || a.variant256 == 4) MT = a.variant256;
    a.mt = ceil_div(a.M, 64 * MT);
    long tiles;
    if ((a.nt & 7) == 0) tiles = (long)a.mt * a.nt;
    else tiles = (long)((a.mt + 7) / 8) * 8 * a.nt;
    // automatic = through LDS: interleaved, order-robust timing (tools/gemm_ab.py) has it at -13 % on qkv (5 rounds of
    // tiles), level on fc1 / fc2 / proj; splitting the rows into whole rounds + a tail launch was measured too and does
    // not add to it
    if (a.direct_store == 2) a.direct_store = 0;
    const dim3 grid((unsigned)tiles), block(G2_THREADS);
    const size_t lds = 256 * 528;      // 2 stages x 64 KiB of ring; the epilogue re-uses it as a 256 x 528 B output tile (132 KiB)
    static bool attr_set = false;
    if (!attr_set) {
#define SETATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        SETATTR(EPI_BIAS); SETATTR(EPI_GELU); SETATTR(EPI_QUICK_GELU); SETATTR(EPI_RESIDUAL); SETATTR(EPI_EMBED);
#undef SETATTR
        attr_set = true;
    }
#define L(E) do { if (MT == 4) VLLM_LAUNCH((gemm256_bf16_kernel<E, 4>), grid, block, lds, st, a); \
                  else VLLM_LAUNCH((gemm256_bf16_kernel<E, 3>), grid, block, lds, st, a); } while (0)
    switch (epi) {
    case EPI_BIAS: L(EPI_BIAS); break;
    case EPI_GELU: L(EPI_GELU); break;
    case EPI_QUICK_GELU: L(EPI_QUICK_GELU); break;
    case EPI_RESIDUAL: L(EPI_RESIDUAL); break;
    case EPI_EMBED: L(EPI_EMBED); break;
    case EPI_F32: L(EPI_F32); break;
    default: set_error("gemm256: unknown epilogue %d", epi); return VLLM_EINVAL;
    }
#undef L
    VLLM_CHECK_LAUNCH("gemm256_bf16_kernel");
    return VLLM_OK;
}

}  // namespace vllm
