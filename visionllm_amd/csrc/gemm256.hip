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
#include <stdlib.h>
#include <type_traits>
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_g2_prof[8];   // VLLM_GEMM_PROF: shader-clock ticks of wave 0 per phase, summed over blocks
constexpr int G2_BN = 256, G2_BK = 64;   // block rows are 64 * MT (template parameter)
constexpr int G2_THREADS = 512;
constexpr int G2_HALF = 128 * G2_BK * 2;          // 16 KiB half-tile
constexpr int G2_STAGE = 4 * G2_HALF;             // A0 A1 B0 B1
constexpr int OFF_A0 = 0, OFF_A1 = G2_HALF, OFF_B0 = 2 * G2_HALF, OFF_B1 = 3 * G2_HALF;

// Refill one half-tile of NSEG x 8 rows (NSEG <= 16): one LDS-DMA instruction per 8 rows, EXACTLY 2 per wave (the counted
// vmcnt waits assume that): waves whose segments do not exist (NSEG < 16) reload the last real segment into the unused
// tail of the 16 KiB slot.
// SWZ1: chunk swizzle (r >> 1) & 7 instead of r & 7.  The 32x32x16 fragments are read as (row = lane & 31, chunk pair
// bit = lane >> 5): a ds_read_b128 lane group then holds rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of ONE chunk, and
// rows 8 apart must land on different banks (same scheme as the attention K tile, attn.hip swz_k).
template <int NSEG, bool SWZ1 = false>
__device__ __forceinline__ void issue_half(const uint16_t *__restrict__ src, int ld, int row0, int nrows, int k0,
                                           char *lds_half, int wave, int lane, int skipP)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int seg = wave * 2 + s;
        const int sseg = seg < NSEG ? seg : NSEG - 1;      // source segment (dummy reload for idle slots)
        const int r = sseg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (SWZ1 ? ((r >> 1) & 7) : (r & 7));
        int grow = row0 + r;
        grow = grow < nrows ? grow : nrows - 1;
        if (skipP > 0) grow += grow / skipP + 1;
        const uint16_t *g = src + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(lds_half + seg * 1024), 16, 0, 0);
    }
}

// sum over the 32 lanes of a half wave (lanes 0-31 / 32-63), in every lane: quad butterfly and two row rotations by DPP, one
// ds_bpermute for the other row of the half
__device__ __forceinline__ float half_wave_sum(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xb1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4e>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x124>{});   // row_ror:4
    v += dpp(v, std::integral_constant<int, 0x128>{});   // row_ror:8
    return v + __shfl_xor(v, 16);
}

#define G2_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G2_BARRIER()                      \
    do {                                  \
        __builtin_amdgcn_s_barrier();     \
        __builtin_amdgcn_sched_barrier(0);\
    } while (0)

// MT = 16-row m tiles per wave per A half: 4 -> 256-row block tile, 3 -> 192 rows (better tile-count quantisation on
// 256 CUs for some shapes: 23080 x 1024 is 364 tiles = 1.42 rounds at 256 rows but 484 = 1.89 rounds at 192).
// MF32 ("gemm_variant" 4 / VLLM_GEMM_FORCE_MF32; opt-in, NOT the default): the same pipeline on
// v_mfma_f32_32x32x16_bf16 (MT = 4 only).  Measured (tools/gemm_ab.py, tools/pmc_gemm_variants.sh): 114 us at 4096^3
// against 100 us for the 16x16x32 schedule and 89 us for hipBLASLt; 165 / 219 us on qkv / fc1 against 146 / 197.  The
// MFMA section of a phase shrinks to 256 cycles but the phase period stays ~470: what paces it is the refill -- two
// LDS-DMA instructions per wave per phase stall the issuing wave ~100 cycles each wherever they are placed (read section
// or between the MFMAs), i.e. the L2 -> LDS path at 64 KiB per 2048 MFMA cycles per CU, not LDS reads (balancing the
// fragment reads 8/4/8/8 per phase changed nothing, SQ_LDS_BANK_CONFLICT is 0.3 % of LDS cycles).  Kept as a measured
// variant; a faster GEMM on this part needs fewer refill bytes per flop (a larger block tile), not a faster inner loop.  The 16x16x32 instruction issues every ~24 cycles
// (2/3 of the matrix peak, tools/probes/mfma_rate.hip), the 32x32x16 one every 32 cycles for twice the flops.  A wave's
// quadrant piece is then two 32(m) x 32(n) blocks, 8 MFMAs per phase; fragments are 16 bytes of one row per lane
// (row = lane & 31, k half = lane >> 5), the k16 step ks selects chunk pair 2ks, 2ks+1 -> byte offset ^ (ks << 5).
// LNC: the consumer side of a folded norm (GemmArgs::ln_in) -- its own instantiation: the epilogue's extra operands must not
// cost the plain kernels a register
template <int EPI, int MT, bool MF32 = false, bool LNC = false>
__global__ __launch_bounds__(G2_THREADS, 1) void gemm256_bf16_kernel(const GemmArgs a0_)
{
    static_assert(!MF32 || MT == 4, "32x32x16 path: 256-row tiles only");
    static_assert(!LNC || !MF32, "folded norm: 16x16x32 schedule only");
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 64 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int wr = wave >> 2, wc = wave & 3;
    const int nk_all = a0_.K / G2_BK;

    // ---- work of this block: one whole tile, or (stream-K tail, see the launcher) up to two K segments of adjacent tiles ----
    // The K iterations of the sk_tiles last tiles form ONE list of sk_tiles * nk entries that sk_blocks blocks share evenly:
    // block `rid` owns [rid * total / sk_blocks, (rid + 1) * total / sk_blocks) -- fewer entries than a tile has, so at most
    // the tail of one tile and the head of the next.  A segment that does not reach its tile's last K iteration leaves its raw
    // fp32 accumulators in scratch slot `rid` (PARTIAL); the block that owns the last iteration of a tile adds the slots of
    // everyone in front of it, in ascending order (deterministic), and runs the epilogue.  A block does its PARTIAL segment
    // first: nobody waits for a block that is itself waiting.
    int seg_t0 = -1, seg_t1 = -1, seg_k0a = 0, seg_k0b = 0, seg_na = nk_all, seg_nb = 0, nseg = 1, rid = 0;
    if (a0_.sk_tiles > 0) {
        if ((int)blockIdx.x < a0_.sk_dp) {
            seg_t0 = blockIdx.x;
        } else {
            const int sb = blockIdx.x - a0_.sk_dp;
            // rid = block index: a finishing block only ever waits for segments of LOWER block indices, which the dispatcher starts
            // first -- forward progress then never needs all stream-K blocks to be resident at once (the round-3 order,
            // list neighbours on one XCD, let block 1 wait for block 248; under CU masking or a busy device that can deadlock).
            rid = sb;
            const long total = (long)a0_.sk_tiles * nk_all;
            const int it0 = (int)(rid * total / a0_.sk_blocks), it1 = (int)((rid + 1) * total / a0_.sk_blocks);
            if (it1 <= it0) return;
            const int tlo = it0 / nk_all, thi = (it1 - 1) / nk_all;
            if (tlo == thi) {
                seg_t0 = a0_.sk_dp + tlo; seg_k0a = it0 - tlo * nk_all; seg_na = it1 - it0;
            } else {
                nseg = 2;
                seg_t0 = a0_.sk_dp + thi; seg_k0a = 0; seg_na = it1 - thi * nk_all;                        // head of the next tile: PARTIAL
                seg_t1 = a0_.sk_dp + tlo; seg_k0b = it0 - tlo * nk_all; seg_nb = nk_all - seg_k0b;         // tail of this tile: finishes it
            }
        }
    }
    for (int seg = 0; seg < nseg; ++seg) {
    // (the lane index is laundered per segment: nothing lane-dependent is loop-invariant, or the compiler hoists the per-lane
    //  address arithmetic of prologue AND epilogue in front of the loop and spills it across the main loop)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int fr = lane & 15, kq = lane >> 4;
    const int tid_l = wave * 64 + lane;
    GemmArgs al = a0_;            // (block-local copy: res_init may be switched off for this block / segment, see below)
    const GemmArgs &a = al;
    const int tile_d = seg == 0 ? seg_t0 : seg_t1;
    const int kbeg = seg == 0 ? seg_k0a : seg_k0b;
    const int nk = seg == 0 ? seg_na : seg_nb;
    const bool finishing = kbeg + nk == nk_all;
    if (!finishing) al.res_init = 0;      // the residual (as the accumulators' initial value) enters once, in the finishing segment

    // ---- XCD-aware tile mapping (as gemm.hip) ----
    int tm_idx, tn_idx;
    if (a.sk_tiles > 0) {
        // dense order (no padding tiles): complete groups of 8 X panels first, mapped as below; then the last mt % 8 panels
        if ((a.nt & 7) == 0) {
            const int xcd = tile_d & 7, s = tile_d >> 3, npx = a.nt >> 3;
            tn_idx = xcd + 8 * (s % npx);
            tm_idx = s / npx;
        } else {
            const int fullp = (a.mt >> 3) * 8 * a.nt;
            if (tile_d < fullp) {
                const int xcd = tile_d & 7, s = tile_d >> 3;
                tm_idx = xcd + 8 * (s / a.nt);
                tn_idx = s % a.nt;
            } else {
                const int vx = a.mt & 7, e = tile_d - fullp;
                tm_idx = (a.mt & ~7) + e % vx;
                tn_idx = e / vx;
            }
        }
    } else {
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
    const unsigned t_start = a.prof ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const unsigned long long rt_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // Residual as accumulator init (below) with a LayerScale needs 1 / ls: every block checks ITS 256 columns (finite, |ls| >=
    // 1e-4: the division must not blow the fp32 accumulation up) and keeps the plain epilogue otherwise -- block-uniform.
    if (EPI == EPI_RESIDUAL && al.res_init && al.scale) {
        const int t = tid_l;
        bool ok = true;
        if (t < G2_BN && n0 + t < al.N) {
            const float v = bf16_to_f32(al.scale[n0 + t]);
            ok = fabsf(v) >= 1e-4f && fabsf(v) < 1e30f;
        }
        if (!__syncthreads_and(ok ? 1 : 0)) al.res_init = 0;
    }
    f32x4_t acc[MF32 ? 1 : 4][2][MT];   // [quadrant q = 2*i + j][n tile][m tile]
    f32x16_t acc32[MF32 ? 4 : 1][2];     // MF32: [quadrant][m block of 32]
    if constexpr (MF32) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[q][j][r] = 0.f;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[q][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // Residual epilogue without LayerScale (the CLIP encoder's proj / fc2): the residual tile becomes the INITIAL VALUE of the
        // accumulators -- res + (X W^T + bias) = (res + X W^T) + bias in fp32, one rounding at the end as before.  A lane owns 4
        // features of one token, so reading the tile in the accumulator layout is 32 eight-byte loads per lane that touch 16 rows x
        // 32 bytes each: 18-24 K cycles per tile wherever they are issued (phase clock: epilogue 7.6 K -> 24-30 K; as plain loads
        // in front of the prologue the same 18-24 K).  Here the tile comes in row-wise instead -- LDS-DMA, 16 bytes per lane, 512
        // contiguous bytes per row, into the (still empty) ring, 16-byte chunk c of row r at chunk c ^ (r & 31) -- and the
        // lanes pick their elements up with conflict-free ds_read_b64.
        if (EPI == EPI_RESIDUAL && a.res_init) {
            constexpr int RROWS = 64 * MT;                       // rows of the block tile
#pragma unroll
            for (int s = 0; s < RROWS / 16; ++s) {               // 8 waves x 2 rows x 512 B per round
                const int r = s * 16 + wave * 2 + (lane >> 5);
                const int pp = lane & 31;                        // chunk position in the LDS row
                const int c = pp ^ (r & 31);                     // source chunk that must land there
                int gm = m0 + r;
                gm = gm < a.M ? gm : a.M - 1;
                int gn = n0 + c * 8;
                gn = gn < a.N ? gn : a.N - 8;                    // (N % 8 == 0 on this route: launcher)
                const uint16_t *g = a.res + (size_t)gm * a.ldr + gn;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(smem + (s * 16 + wave * 2) * 512), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int qi = q >> 1, qj = q & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int nl = qj * 128 + wc * 32 + i * 16 + kq * 4;   // column of the tile: chunk nl / 8, half (nl / 4) & 1
                    float il[4] = {1.f, 1.f, 1.f, 1.f};                     // 1 / LayerScale of this lane's 4 features
                    if (a.scale && n0 + nl < a.N) {
                        const uint2_t sv = *reinterpret_cast<const uint2_t *>(a.scale + n0 + nl);
                        il[0] = __builtin_amdgcn_rcpf(bf16lo_to_f32(sv.x)); il[1] = __builtin_amdgcn_rcpf(bf16hi_to_f32(sv.x));   // (1 ulp: the round trip
                        il[2] = __builtin_amdgcn_rcpf(bf16lo_to_f32(sv.y)); il[3] = __builtin_amdgcn_rcpf(bf16hi_to_f32(sv.y));   //  res / ls * ls costs 2^-23 |res|)
                    }
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        const int ml = qi * (32 * MT) + wr * (16 * MT) + j * 16 + fr;
                        const uint2_t rr = *reinterpret_cast<const uint2_t *>(smem + ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + ((nl >> 2) & 1) * 8);
                        acc[q][i][j] = (f32x4_t){bf16lo_to_f32(rr.x) * il[0], bf16hi_to_f32(rr.x) * il[1], bf16lo_to_f32(rr.y) * il[2], bf16hi_to_f32(rr.y) * il[3]};
                    }
                }
            }
            G2_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();                        // the ring is free again: the prologue's DMA may overwrite it
        }
    }

    // per-lane LDS byte offsets inside a half-tile for the two fragment kinds (ks = 0 / 1 differ by XOR 4 chunks)
    int xoff[MT], woff[2];
    const int l31 = lane & 31, hi = lane >> 5;
    if constexpr (MF32) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = wr * 64 + t * 32 + l31;
            xoff[t] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
        }
        const int r = wc * 32 + l31;
        woff[0] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);
    } else {
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
    }
    bf16x8_t wg[2][2];              // second W fragment set: B0's fragments (wf) stay live from phase 1 to phase 4
    bf16x8_t xf[MT][2], wf[2][2];   // MF32: xf[2 m blocks x 2][..] viewed as [jb*2 + ks/2][ks&1], wf[ks/2][ks&1]
    bf16x8_t xg[MF32 ? 2 : 1][2];   // MF32: third 16-register X buffer (see the MF32 loop)

    auto k_of = [&](int t) { return (kbeg + (t < nk ? t : nk - 1)) * G2_BK; };   // clamped: tail refills are harmless
    auto issue_A = [&](int half, int stage, int t) {
        issue_half<4 * MT, MF32>(a.X, a.ldx, m0 + half * (32 * MT), a.M, k_of(t), smem + stage * G2_STAGE + (half ? OFF_A1 : OFF_A0),
                           wave, lane, LNC ? 0 : a.xP);   // (a folded norm never feeds the CLS-skipping loader: no division to carry)
    };
    auto issue_B = [&](int half, int stage, int t) {
        issue_half<16, MF32>(a.W, a.ldw, n0 + half * 128, a.N, k_of(t), smem + stage * G2_STAGE + (half ? OFF_B1 : OFF_B0), wave,
                       lane, 0);
    };
    auto read_x = [&](const char *half) {
        if constexpr (MF32) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    xf[jb * 2 + (ks >> 1)][ks & 1] = *reinterpret_cast<const bf16x8_t *>(half + (xoff[jb] ^ (ks << 5)));
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                xf[t][0] = *reinterpret_cast<const bf16x8_t *>(half + xoff[t]);
                xf[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (xoff[t] ^ 64));   // chunk index + 4  (ks = 1)
            }
        }
    };
    auto read_w_into = [&](const char *half, bf16x8_t (&dst)[2][2]) {
        if constexpr (MF32) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dst[ks >> 1][ks & 1] = *reinterpret_cast<const bf16x8_t *>(half + (woff[0] ^ (ks << 5)));
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                dst[t][0] = *reinterpret_cast<const bf16x8_t *>(half + woff[t]);
                dst[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (woff[t] ^ 64));
            }
        }
    };
    auto read_w = [&](const char *half) { read_w_into(half, wf); };
#define G2_MMA(Q, MID, WF)                                                                                      \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        if constexpr (MF32) {                                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                  \
                _Pragma("unroll") for (int jb = 0; jb < 2; ++jb)                                                \
                    acc32[Q][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks >> 1][ks & 1], xf[jb * 2 + (ks >> 1)][ks & 1], acc32[Q][jb], 0, 0, 0); \
                if (ks == 0) { MID; }   /* the refill's LDS-DMA goes out under the first MFMAs of the phase */      \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
                    _Pragma("unroll") for (int j = 0; j < MT; ++j)                                              \
                        acc[Q][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i][ks], xf[j][ks], acc[Q][i][j], 0, 0, 0); \
        }                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

#define G2_MMA32(Q, XB0, O0, XB1, O1, MID)                                                                      \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                      \
            acc32[Q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks >> 1][ks & 1], XB0[O0 + (ks >> 1)][ks & 1], acc32[Q][0], 0, 0, 0); \
            acc32[Q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks >> 1][ks & 1], XB1[O1 + (ks >> 1)][ks & 1], acc32[Q][1], 0, 0, 0); \
            if (ks == 0) { MID; }                                                                               \
        }                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    // Folded norm, consumer side (LNC; rows of exactly four column tiles, i.e. hidden size 1024): the tile's 256 x 4 x {mean, M2}
    // (8 KiB, contiguous) and its 256 column sums / fp32 biases come in by LDS-DMA, IN FRONT of the prologue's DMA (older than all
    // of it: the counted vmcnt waits retire them first, no register is held, no wait of their own), into LDS behind the ring and the
    // epilogue's output tile; in front of the epilogue's first barrier 256 threads turn the statistics into {r, -r * mean} per
    // row (Chan's update in a fixed order: exact block means, no cancellation).
    constexpr int LN_TAB = 256 * 528, LN_CS = LN_TAB + 2048, LN_RAW = LN_TAB + 4096;   // byte offsets in LDS
    if (LNC && finishing) {
        {   // wave w: rows 32 w .. 32 w + 31 (two lanes per row, 16 bytes each)
            int row = m0 + wave * 32 + (lane >> 1);
            row = row < a.M ? row : a.M - 1;
            const float *src = a.ln_in + (size_t)row * 8 + (lane & 1) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(smem + LN_RAW + wave * 1024), 16, 0, 0);
        }
        if (wave < 2) {   // 256 floats = one 1 KiB instruction each: wave 0 the column sums, wave 1 the biases (a missing vector: zeros below)
            const float *vec = wave == 0 ? a.ln_colsum : a.ln_bias;
            int nn = n0 + lane * 4;
            nn = nn + 4 <= a.N ? nn : a.N - 4;
            if (vec) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vec + nn),
                                                      (__attribute__((address_space(3))) void *)(smem + LN_CS + wave * 1024), 16, 0, 0);
        }
    }
    // ---- prologue: tile 0 complete in stage 0; tile 1's A0, B1, A1 in flight in stage 1 ----
    issue_A(0, 0, 0); issue_B(0, 0, 0); issue_B(1, 0, 0); issue_A(1, 0, 0);
    issue_A(0, 1, 1); issue_B(1, 1, 1); issue_A(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G2_BARRIER();
    const unsigned t_pro = a.prof ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    if (wr == 1) G2_BARRIER();   // stagger: group 1 runs one barrier behind group 0

    if constexpr (MF32) {
        // 32x32x16 schedule.  A phase's MFMA section is 8 x 32 = 256 cycles, so the other group's read section has to fit
        // in that: (a) the refill's LDS-DMA is issued INSIDE the MFMA section (after the first two MFMAs), (b) the twelve
        // fragment reads of phase 1 are split: the first m block of A0(t+1) is read one phase early, in phase 4 of tile t
        // (A0(t+1) is retired by a counted vmcnt(8) + the barrier of phase 3) -> reads per phase 8 / 4 / 8 / 8.
        // X fragment buffers (16 registers each): R0 = xf[0..1], R1 = xf[2..3], R2 = xg[0..1].
        //   A0: m block 0 in R0, m block 1 in R1 (phases 1, 2);  A1: m block 0 in R1, m block 1 in R2 (phases 3, 4).
        auto read_blk = [&](const char *half, int jb, bf16x8_t (&dst)[2][2]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                dst[ks >> 1][ks & 1] = *reinterpret_cast<const bf16x8_t *>(half + (xoff[jb] ^ (ks << 5)));
        };
        bf16x8_t (&R0)[2][2] = *reinterpret_cast<bf16x8_t (*)[2][2]>(&xf[0]);
        bf16x8_t (&R1)[2][2] = *reinterpret_cast<bf16x8_t (*)[2][2]>(&xf[2]);
        read_blk(smem + OFF_A0, 0, R0);   // tile 0's first block (later tiles: phase 4 of the tile before)
        for (int t = 0; t < nk; ++t) {
            const int s = t & 1;
            const char *st = smem + s * G2_STAGE;
            // phase 1: quadrant (A0,B0)
            read_blk(st + OFF_A0, 1, R1); read_w(st + OFF_B0);
            G2_WAIT_LGKM0(); G2_BARRIER();
            G2_MMA32(0, xf, 0, xf, 2, issue_B(0, s ^ 1, t + 1));
            G2_BARRIER();
            // phase 2: quadrant (A0,B1)
            read_w(st + OFF_B1);
            G2_WAIT_LGKM0(); G2_BARRIER();
            G2_MMA32(1, xf, 0, xf, 2, issue_A(0, s, t + 2));
            G2_BARRIER();
            // phase 3: quadrant (A1,B1); retire A0(t+1) for the early read of phase 4
            read_blk(st + OFF_A1, 0, R1); read_blk(st + OFF_A1, 1, xg);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            G2_WAIT_LGKM0(); G2_BARRIER();
            G2_MMA32(3, xf, 2, xg, 0, issue_B(1, s, t + 2));
            G2_BARRIER();
            // phase 4: quadrant (A1,B0); first block of A0(t+1) (other stage) read early
            read_w(st + OFF_B0);
            read_blk(smem + (s ^ 1) * G2_STAGE + OFF_A0, 0, R0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            G2_WAIT_LGKM0(); G2_BARRIER();
            G2_MMA32(2, xf, 2, xg, 0, issue_A(1, s, t + 2));
            G2_BARRIER();
        }
    } else {
    for (int t = 0; t < nk; ++t) {
        const int s = t & 1;
        const char *st = smem + s * G2_STAGE;
        // phase 1: quadrant (A0,B0); refill B0 of the OTHER stage with tile t+1
        read_x(st + OFF_A0); read_w(st + OFF_B0);
        if constexpr (!MF32) issue_B(0, s ^ 1, t + 1);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(0, issue_B(0, s ^ 1, t + 1), wf);
        G2_BARRIER();
        // phase 2: quadrant (A0,B1); refill A0 (this stage) with tile t+2
        read_w_into(st + OFF_B1, wg);   // B0's fragments stay in wf for phase 4 (4 fewer ds_read_b128 per K tile)
        if constexpr (!MF32) issue_A(0, s, t + 2);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(1, issue_A(0, s, t + 2), wg);
        G2_BARRIER();
        // phase 3: quadrant (A1,B1); refill B1 with tile t+2
        read_x(st + OFF_A1);
        if constexpr (!MF32) issue_B(1, s, t + 2);
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(3, issue_B(1, s, t + 2), wg);
        G2_BARRIER();
        // phase 4: quadrant (A1,B0); refill A1 with tile t+2; retire everything but the last 3 half-tiles (MF32: the refill
        // of this phase is issued after the wait, inside the MFMA section -> 2 half-tiles outstanding at the wait)
        if constexpr (!MF32) {
            issue_A(1, s, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        G2_WAIT_LGKM0(); G2_BARRIER();
        G2_MMA(2, issue_A(1, s, t + 2), wf);
        G2_BARRIER();
    }
    }
    if (wr == 0) G2_BARRIER();   // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
    const unsigned t_loop = a.prof ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    if (a.sk_tiles > 0 && (int)blockIdx.x >= a.sk_dp) {
        // Hand-off between workgroups (cdna_hip_programming.md section 6, guideline 16, form R1): the payload is stored
        // WRITE-THROUGH (sc1, 16 bytes per lane) and read back with sc1 loads, so neither side needs a cache-wide fence and
        // nothing depends on which XCD a block runs on; every storing wave drains its stores, ONE lane publishes the flag
        // (relaxed, agent scope); the reader polls that one word relaxed.  (A first version with __threadfence() in every
        // thread and acquire polls cost ~100 us per GEMM: each is a walk over the XCD's L2.)
        // scratch slot: 32 16-byte vectors per thread, vector v of thread t at (v * 512 + t) -> every access is a coalesced 8 KiB row
        const __amdgpu_buffer_rsrc_t ws_rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.sk_ws, 0, (int)(a.sk_blocks * (32 * G2_THREADS * 16)), 0x00020000);
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        constexpr int AUX_SC1 = 16;
        if (!finishing) {
            const int base = (rid * (32 * G2_THREADS) + tid_l) * 16;
            if constexpr (MF32) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4_t v = {acc32[q][jb][4 * g], acc32[q][jb][4 * g + 1], acc32[q][jb][4 * g + 2], acc32[q][jb][4 * g + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws_rs, base + ((q * 2 + jb) * 4 + g) * (G2_THREADS * 16), 0, AUX_SC1);
                        }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[q][i][j]), ws_rs, base + ((q * 2 + i) * MT + j) * (G2_THREADS * 16), 0, AUX_SC1);
            }
            asm volatile("s_waitcnt vmcnt(0) ; stream-K publish: every storing wave drains" ::: "memory");
            __syncthreads();
            if (tid_l == 0) __hip_atomic_store(a.sk_flags + rid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // finishing: the segments in front of this one, lowest K first
        const long total = (long)a.sk_tiles * nk_all;
        const long ts = (long)(tile_d - a.sk_dp) * nk_all;
        int c = (int)(ts * a.sk_blocks / total);
        while ((long)(c + 1) * total / a.sk_blocks <= ts) ++c;
        for (; c < rid; ++c) {
            if (tid_l == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(a.sk_flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 26)) __builtin_trap();   // (minutes: the producer is gone -- fail loudly instead of hanging the device)
                }
            }
            __syncthreads();
            const int base = (c * (32 * G2_THREADS) + tid_l) * 16;
            // a quadrant (8 vectors = 32 registers) at a time: all 32 loads in flight at once would not fit beside the accumulators
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (MF32) {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4_t p = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ws_rs, base + ((q * 2 + jb) * 4 + g) * (G2_THREADS * 16), 0, AUX_SC1));
                            acc32[q][jb][4 * g] += p[0]; acc32[q][jb][4 * g + 1] += p[1]; acc32[q][jb][4 * g + 2] += p[2]; acc32[q][jb][4 * g + 3] += p[3];
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
                            acc[q][i][j] += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ws_rs, base + ((q * 2 + i) * MT + j) * (G2_THREADS * 16), 0, AUX_SC1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                 // everyone has read slot c: its flag goes back to zero for the next launch
            if (tid_l == 0) __hip_atomic_store(a.sk_flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    struct ProfEnd {
        const GemmArgs &a; unsigned t0, t1, t2; int w, l; unsigned long long rt0;
        __device__ ~ProfEnd() {
            if (a.trace && w == 0 && l == 0 && blockIdx.x < 8192) {
                a.trace[blockIdx.x * 3 + 0] = rt0;
                a.trace[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
                a.trace[blockIdx.x * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
            }
            if (a.prof && w == 0 && l == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stores have been accepted by the memory system)
                const unsigned t3 = (unsigned)__builtin_amdgcn_s_memtime();
                atomicAdd(&g_g2_prof[0], (unsigned long long)(t1 - t0)); atomicAdd(&g_g2_prof[1], (unsigned long long)(t2 - t1));
                atomicAdd(&g_g2_prof[2], (unsigned long long)(t3 - t2)); atomicAdd(&g_g2_prof[3], 1ull);
            }
        }
    } prof_end{a, t_start, t_pro, t_loop, wave, lane, rt_start};

    // ---- epilogue ----
    // bf16 outputs leave through LDS: the accumulator layout gives a wave store of 16 rows x 32 contiguous bytes
    // (11.1 us per 32 MB round of tiles, tools/probes/store_pattern.hip); re-read row-wise the tile leaves as 16-byte
    // lane stores, 512 contiguous bytes per row (7.1 us).  The pipeline's LDS is free by now; rows are padded to 528 B.
    constexpr int OPITCH = 528;
    const bool via_lds = !a.direct_store && EPI != EPI_F32 && (a.N & 7) == 0 && (a.ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15u) == 0;
    if (via_lds) {
        if (LNC) {   // the row table of the folded norm, from the statistics that came in with the prologue (here, not there: the
                     // prologue has no register to spare, and a spill reload between its DMA issues drains them -- 5 K ticks)
            if (tid_l < BM_) {
                const float2_t *sp = reinterpret_cast<const float2_t *>(smem + LN_RAW) + tid_l * 4;
                float r_, nrm_;
                if (a.ln_rms) {
                    const float ss = (sp[0].x + sp[1].x) + (sp[2].x + sp[3].x);
                    r_ = __builtin_amdgcn_rsqf(fmaf(ss, a.ln_inv_cols, a.ln_eps));
                    nrm_ = 0.f;
                } else {
                    float mean = sp[0].x, m2 = sp[0].y;   // (the launcher's constants: ln_cw[0] = 1, ln_cc[0] = 0)
#pragma unroll
                    for (int sidx = 1; sidx < 4; ++sidx) {
                        const float2_t st_ = sp[sidx];
                        const float dlt = st_.x - mean;
                        mean = fmaf(dlt, a.ln_cw[sidx], mean);
                        m2 += fmaf(dlt * dlt, a.ln_cc[sidx], st_.y);
                    }
                    r_ = __builtin_amdgcn_rsqf(fmaf(m2, a.ln_inv_cols, a.ln_eps));
                    nrm_ = -r_ * mean;
                }
                reinterpret_cast<float2_t *>(smem + LN_TAB)[tid_l] = (float2_t){r_, nrm_};
            }
            if (tid_l < 2 * G2_BN && !(tid_l < G2_BN ? a.ln_colsum : a.ln_bias)) reinterpret_cast<float *>(smem + LN_CS)[tid_l] = 0.f;
        }
        __builtin_amdgcn_s_barrier();   // every wave is out of the main loop: no fragment read of the ring is pending
        const float2_t *ln_tab = reinterpret_cast<const float2_t *>(smem + 256 * OPITCH);               // (= LN_TAB, just built)
        const float *ln_cs = reinterpret_cast<const float *>(smem + 256 * OPITCH + 2048), *ln_bs = ln_cs + G2_BN;
        if constexpr (MF32) {
            // accumulator register 4g + e of block jb: feature wc*32 + 8g + 4hi + e, token wr*64 + jb*32 + l31
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int qi = q >> 1, qj = q & 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = qj * 128 + wc * 32 + g * 8 + hi * 4;
                    const int n = n0 + nl;
                    const bool nok = n < a.N;
                    const EpiCols cols = epi_cols<EPI>(a, nok ? n : 0);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const int ml = qi * 128 + wr * 64 + jb * 32 + l31;
                        const int m = m0 + ml;
                        const f32x4_t av = {acc32[q][jb][4 * g], acc32[q][jb][4 * g + 1], acc32[q][jb][4 * g + 2], acc32[q][jb][4 * g + 3]};
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (nok && m < a.M) epi_value<EPI>(a, m, n, av, cols, v);
                        uint2_t o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2_t *>(smem + ml * OPITCH + nl * 2) = o;
                    }
                }
            }
        } else {
        float2_t rn[LNC ? 2 : 1][LNC ? MT : 1];   // folded norm: {r, -r mean} of this lane's 2 x MT rows (the fragment registers are free by now)
        if constexpr (LNC) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int j = 0; j < MT; ++j) rn[qi][j] = ln_tab[qi * (32 * MT) + wr * (16 * MT) + j * 16 + fr];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qi = q >> 1, qj = q & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = qj * 128 + wc * 32 + i * 16 + kq * 4;
                const int n = n0 + nl;
                const bool nok = n < a.N;
                EpiCols cols = epi_cols<EPI>(a, nok ? n : 0);
                f32x4_t csum = {0.f, 0.f, 0.f, 0.f};
                if (LNC) {   // folded norm: fp32 bias' and column sums of the gamma-scaled weight (staged in the prologue)
                    const f32x4_t b4 = *reinterpret_cast<const f32x4_t *>(ln_bs + nl);
                    cols.bia[0] = b4[0]; cols.bia[1] = b4[1]; cols.bia[2] = b4[2]; cols.bia[3] = b4[3];
                    csum = *reinterpret_cast<const f32x4_t *>(ln_cs + nl);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int ml = qi * (32 * MT) + wr * (16 * MT) + j * 16 + fr;
                    const int m = m0 + ml;
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
                    if (nok && m < a.M) {
                        if constexpr (LNC) {
                            epi_value_folded<EPI>(acc[q][i][j], rn[qi][j].x, rn[qi][j].y, csum, cols, v);
                        } else {
                            epi_value<EPI>(a, m, n, acc[q][i][j], cols, v);
                        }
                    }
                    uint2_t o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2_t *>(smem + ml * OPITCH + nl * 2) = o;
                }
            }
        }
        }
        __builtin_amdgcn_s_barrier();
        const int t = tid_l, c8 = (t & 31) * 8;   // 32 lanes x 16 B = one 512-byte tile row
#pragma unroll 4
        for (int p = 0; p < BM_ / 16; ++p) {
            const int ml = p * 16 + (t >> 5), m = m0 + ml, n = n0 + c8;
            const bool live = m < a.M && n < a.N;
            uint4_t o = {0, 0, 0, 0};
            if (live) {
                o = *reinterpret_cast<const uint4_t *>(smem + ml * OPITCH + c8 * 2);
                *reinterpret_cast<uint4_t *>(a.Y + epi_out_row<EPI>(a, m) * a.ldy + n) = o;
            }
            if (a.ln_out) {
                // folded norm, producer side: statistics of the 256 (or fewer: last column tile) bf16 values of this row that were just
                // stored -- a row is the 32 lanes of a half wave: mean, then the centred second moment (two passes over registers)
                const uint32_t u[4] = {o.x, o.y, o.z, o.w};
                float x_[8];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) { x_[2 * k2] = bf16lo_to_f32(u[k2]); x_[2 * k2 + 1] = bf16hi_to_f32(u[k2]); }
                const float nb = (float)min(G2_BN, a.N - n0);
                float s0, s1;
                if (a.ln_rms) {
                    float q2 = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) q2 = fmaf(x_[k2], x_[k2], q2);
                    s0 = half_wave_sum(q2); s1 = 0.f;
                } else {
                    float sum = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) sum += x_[k2];
                    s0 = half_wave_sum(sum) / nb;
                    float q2 = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) { const float d = live ? x_[k2] - s0 : 0.f; q2 = fmaf(d, d, q2); }
                    s1 = half_wave_sum(q2);
                }
                if ((t & 31) == 0 && m < a.M)
                    *reinterpret_cast<float2_t *>(a.ln_out + ((size_t)m * a.nt + tn_idx) * 2) = (float2_t){s0, s1};
            }
        }
        __syncthreads();   // (a second segment's prologue overwrites the ring)
        continue;
    }
    if constexpr (MF32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qi = q >> 1, qj = q & 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + qj * 128 + wc * 32 + g * 8 + hi * 4;
                if (n >= a.N) continue;
                const EpiCols cols = epi_cols<EPI>(a, n);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int m = m0 + qi * 128 + wr * 64 + jb * 32 + l31;
                    if (m >= a.M) continue;
                    const f32x4_t av = {acc32[q][jb][4 * g], acc32[q][jb][4 * g + 1], acc32[q][jb][4 * g + 2], acc32[q][jb][4 * g + 3]};
                    epi_store<EPI>(a, m, n, av, cols);
                }
            }
        }
    } else {
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
    }   // segments
}

static bool res_init_disabled()
{
    static const int v = [] { const char *e = getenv("VLLM_GEMM_RES_INIT"); return e && e[0] == '0' ? 1 : 0; }();   // A/B switch
    return v != 0;
}

bool gemm256p_takes(int epi, const GemmArgs &a, int cus);                                   // gemm256p.hip
int gemm256p_launch(int epi, int MT, const GemmArgs &a, int cus, hipStream_t st);
int gemm256p_debug_counters(long *out, int n);

static long g_sk_launches = 0;   // launches that took the stream-K tail (vllm_gemm_sk_launches: tests assert the path they mean to cover ran)
long gemm256_sk_launches() { return g_sk_launches; }

// VLLM_GEMM_SK=0 switches the stream-K tail off (A/B); VLLM_GEMM_SK_MAX=<tiles> overrides the largest tail that takes it
static bool sk_disabled()
{
    static const int v = [] { const char *e = getenv("VLLM_GEMM_SK"); return e && e[0] == '0' ? 1 : 0; }();
    return v != 0;
}
static double sk_fixup_cost()   // VLLM_GEMM_SK_FIX=<units>: the fix-up's charge in the launcher's cost model (100 = one 256-row K = 1024 tile)
{
    static const double v = [] { const char *e = getenv("VLLM_GEMM_SK_FIX"); return e ? atof(e) : 130.0; }();
    return v;
}
static long sk_max_tiles(int cus)
{
    static const long v = [] { const char *e = getenv("VLLM_GEMM_SK_MAX"); return e ? atol(e) : -1L; }();
    return v >= 0 ? v : (long)cus * 15 / 16;
}

int gemm256_bf16_launch(int epi, GemmArgs a, hipStream_t st)
{
    const int cus = device_cus();
    a.nt = ceil_div(a.N, G2_BN);
    if (a.ln_in || a.ln_out) {   // folded norm: the row-wise LDS epilogue is where it lives
        VLLM_REQUIRE(epi != EPI_F32 && (a.N & 7) == 0 && (a.ldy & 7) == 0 && aligned16(a.Y) && a.variant256 != 5,
                     "gemm256: a folded norm needs the bf16 row-wise epilogue (N, ldy multiples of 8, 16-byte aligned Y, not the 32x32x16 variant)");
        VLLM_REQUIRE(!a.ln_wide || (a.ln_rms && (!a.ln_in || (a.ln_slots >= 1 && a.ln_slots <= 16 && a.ln_cols == a.K && aligned16(a.ln_in) &&
                                                              (!a.ln_bias || aligned16(a.ln_bias)) && (a.N & 3) == 0))),
                     "gemm256: wide folded-norm statistics are RMSNorm sums of squares in at most 16 column tiles per row");
        VLLM_REQUIRE(a.ln_wide || !a.ln_in || (a.ln_slots == 4 && aligned16(a.ln_in) && a.ln_cols == a.K && (a.ln_rms || a.ln_colsum) &&
                                  (!a.ln_colsum || aligned16(a.ln_colsum)) && (!a.ln_bias || aligned16(a.ln_bias)) && (a.N & 3) == 0),
                     "gemm256: folded norm, consumer: statistics of K-element rows in FOUR column tiles (768 < K <= 1024), column sums (LayerNorm), 16-byte aligned fp32 vectors");
        VLLM_REQUIRE(!a.ln_out || (reinterpret_cast<uintptr_t>(a.ln_out) & 7u) == 0, "gemm256: folded norm, producer: misaligned statistics buffer");
        a.direct_store = 0;
        if (a.ln_in && a.ln_wide) a.ln_inv_cols = 1.f / (float)a.ln_cols;
        else if (a.ln_in) {
            float cnt = 0.f;
            for (int sidx = 0; sidx < 4; ++sidx) {
                const float nb = (float)std::min(G2_BN, a.ln_cols - sidx * G2_BN), tot = cnt + nb;
                a.ln_cw[sidx] = nb / tot;
                a.ln_cc[sidx] = cnt * nb / tot;
                cnt = tot;
            }
            a.ln_inv_cols = 1.f / cnt;
        }
    }
    a.res_init = (epi == EPI_RESIDUAL && a.variant256 != 5 && !res_init_disabled() && a.N % 8 == 0 && a.N >= 8 &&
                  a.ldr % 8 == 0 && aligned16(a.res)) ? 1 : 0;
    { static const int pf = [] { const char *e = getenv("VLLM_GEMM_PROF"); return e ? atoi(e) : 0; }(); a.prof = pf; }
    // VLLM_GEMM_TRACE names a raw device address every block writes 24 bytes to: debug builds only (-DVLLM_GEMM_TRACE_ENABLE,
    // tools/trace_gemm256.py builds its own library); the address is checked to be device memory of at least the size the
    // kernel can write before it is trusted.  A release build ignores the variable.
#ifdef VLLM_GEMM_TRACE_ENABLE
    {
        static unsigned long long *const tr = [] {
            const char *e = getenv("VLLM_GEMM_TRACE");
            unsigned long long *p = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr;
            if (p) {
                hipPointerAttribute_t at;
                void *base = nullptr;
                size_t size = 0;
                const bool ok = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice &&
                                hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p) == hipSuccess &&
                                (char *)base + size >= (char *)p + 3 * 8192 * sizeof(unsigned long long);
                if (!ok) {
                    fprintf(stderr, "libvllm_hip: VLLM_GEMM_TRACE=%s is not a device buffer of 3 x 8192 uint64 -- ignored\n", e);
                    p = nullptr;
                }
            }
            return p;
        }();
        a.trace = tr;
    }
#else
    a.trace = nullptr;
#endif
    // block rows 256 (MT=4) or 192 (MT=3): pick the one with the smaller (rounds x tile cost) on this many CUs
    // measured: a 192-row tile costs 0.87 of a 256-row tile (12 instead of 16 MFMAs per phase, same barriers)
    // Stream-K tail (when the caller provides scratch): the tiles of the last, incomplete round do not get a block each -- their
    // K iterations are split evenly over `cus` blocks (kernel comment), so the round costs its share of a tile + the fix-up.
    // The fix-up is NOT small on this part: every block writes one 256 KiB fp32 slot and reads one to three -- 128 MB through
    // the fabric per launch, 30-45 us measured on the ViT-L shapes whatever K is (profiles/r03_gemm_stream_k.txt), i.e. more
    // than a whole K = 1024 tile (33 us).  In units of that tile (100) the fix-up is charged 130, so the tail only takes the
    // stream-K route when a tile is several times longer than that (K >= 4096 and a mostly empty last round).
    // hipBLASLt picks the same macro tile with stream-K for these shapes (profiles/r03_hipblaslt_kernel_names.txt).
    const int nk = a.K / G2_BK;
    const bool sk_possible = a.sk_ws && a.sk_flags && a.sk_ws_bytes >= (long)cus * 32 * G2_THREADS * 16 && (cus & 7) == 0 &&
                             aligned16(a.sk_ws) && !sk_disabled();
    const double sk_fix = sk_fixup_cost();
    auto plan = [&](int mt_rows, int *sk_tiles) -> double {
        const long T = (long)ceil_div(a.M, 64 * mt_rows) * a.nt;
        const long full = T / cus, r = T % cus;
        const double c = (mt_rows == 4 ? 100.0 : 87.0) * nk / 16.0;
        *sk_tiles = 0;
        if (r == 0) return (double)full * c;
        const double dp = (double)(full + 1) * c, sk = ((double)full + (double)r / cus) * c + sk_fix;
        if (sk_possible && r * nk >= cus && r <= sk_max_tiles(cus) && sk < dp) { *sk_tiles = (int)r; return sk; }
        return dp;
    };
    int sk3 = 0, sk4 = 0;
    const double c3 = plan(3, &sk3), c4 = plan(4, &sk4);
    int MT = c3 < c4 ? 3 : 4;
    if (a.variant256 == 3 || a.variant256 == 4) MT = a.variant256;
    const bool mf32 = a.variant256 == 5;   // 256-row tiles on the 32x32x16 instruction
    if (mf32) MT = 4;
    a.mt = ceil_div(a.M, 64 * MT);
    a.sk_tiles = MT == 3 ? sk3 : sk4;
    // qkv / fc1 shapes: the persistent schedule (gemm256p.hip) -- the ring is refilled across tile boundaries, stores drain
    // under the next tile's main loop: ~10 % per tile (tools/gemm_persist_ab.py), more than the stream-K tail recovers, so it
    // is asked first, with the tile height that needs fewer whole rounds
    if (!mf32) {
        // (a last round whose tiles, cut in two, still fit the grid runs as half-height tiles in the persistent kernel.  Measured: 0.9 of a
        //  round, not 0.5 -- a half tile still refills three of the four half-tiles per K tile, and the refills pace the loop)
        auto rounds = [&](int mt_rows) {
            const long T_ = (long)ceil_div(a.M, 64 * mt_rows) * a.nt, full = T_ / cus, r_ = T_ % cus;
            const double c_ = mt_rows == 4 ? 100.0 : 87.0;
            const double last = r_ == 0 ? 0.0 : (gemm_half_tail() && epi != EPI_RESIDUAL && full >= 1 && 2 * r_ <= cus) ? 0.9 : 1.0;
            return ((double)full + last) * c_;
        };
        int pmt = rounds(3) < rounds(4) ? 3 : 4;
        if (a.variant256 == 3 || a.variant256 == 4) pmt = a.variant256;
        GemmArgs b = a;
        b.sk_tiles = 0;
        b.mt = ceil_div(a.M, 64 * pmt);
        if (gemm256p_takes(epi, b, cus)) return a.dry_run ? VLLM_OK : gemm256p_launch(epi, pmt, b, cus, st);
    }
    if (a.dry_run) return VLLM_EINVAL;   // (asked only whether the persistent schedule would take this GEMM: no message, no launch)
    VLLM_REQUIRE(!a.ln_wide, "gemm256: wide folded-norm statistics are implemented by the persistent schedule only (at least as many tiles as CUs, "
                             "aligned operands); the caller launches the norm instead for this shape");
    long tiles;
    if (a.sk_tiles > 0) {
        ++g_sk_launches;
        a.sk_dp = (int)((long)a.mt * a.nt - a.sk_tiles);
        a.sk_blocks = cus;
        tiles = (long)a.sk_dp + cus;
    } else if ((a.nt & 7) == 0) tiles = (long)a.mt * a.nt;
    else tiles = (long)((a.mt + 7) / 8) * 8 * a.nt;
    // automatic = through LDS: interleaved, order-robust timing (tools/gemm_ab.py) has it at -13 % on qkv (5 rounds of
    // tiles), level on fc1 / fc2 / proj; splitting the rows into whole rounds + a tail launch was measured too and does
    // not add to it
    if (a.direct_store == 2) a.direct_store = 0;
    const dim3 grid((unsigned)tiles), block(G2_THREADS);
    const size_t lds = 256 * 528 + 4096 + 8192;   // 2 stages x 64 KiB of ring; the epilogue re-uses it as a 256 x 528 B output tile (132 KiB); behind it 256 x {r, -r mean}, 256 column sums, 256 biases and the 8 KiB of raw statistics of a folded norm
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
#define SETATTR_LN(E) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 3, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        SETATTR_LN(EPI_BIAS); SETATTR_LN(EPI_GELU); SETATTR_LN(EPI_QUICK_GELU);
#undef SETATTR_LN
#define SETATTR(E) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256_bf16_kernel<E, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        SETATTR(EPI_BIAS); SETATTR(EPI_GELU); SETATTR(EPI_QUICK_GELU); SETATTR(EPI_RESIDUAL); SETATTR(EPI_EMBED); SETATTR(EPI_F32);
#undef SETATTR
    }
#define L(E) do { if (mf32) VLLM_LAUNCH((gemm256_bf16_kernel<E, 4, true>), grid, block, lds, st, a); \
                  else if (MT == 4) VLLM_LAUNCH((gemm256_bf16_kernel<E, 4>), grid, block, lds, st, a); \
                  else VLLM_LAUNCH((gemm256_bf16_kernel<E, 3>), grid, block, lds, st, a); } while (0)
    if (a.ln_in) {   // consumer of a folded norm: its own instantiations (bias / GELU / quick-GELU epilogues)
        VLLM_REQUIRE(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_QUICK_GELU, "gemm256: a folded norm feeds a bias / GELU / quick-GELU epilogue");
#define LLN(E) do { if (MT == 4) VLLM_LAUNCH((gemm256_bf16_kernel<E, 4, false, true>), grid, block, lds, st, a); \
                    else VLLM_LAUNCH((gemm256_bf16_kernel<E, 3, false, true>), grid, block, lds, st, a); } while (0)
        if (epi == EPI_BIAS) LLN(EPI_BIAS); else if (epi == EPI_GELU) LLN(EPI_GELU); else LLN(EPI_QUICK_GELU);
#undef LLN
        VLLM_CHECK_LAUNCH("gemm256_bf16_kernel (folded norm)");
        return VLLM_OK;
    }
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

int gemm256_debug_counters(long *out, int n)
{
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_g2_prof), sizeof(h)) != hipSuccess) {
        set_error("gemm256_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 8; ++i) out[i] = (long)h[i];
    const unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_g2_prof), z, sizeof(z));
    if (n >= 12) {   // slots 4-11: the persistent schedule's clock (gemm256p.hip): main-loop ticks, epilogue ticks, tiles, a tile's K tile 0 / 1 / 2 / 3, -
        long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (gemm256p_debug_counters(pp, 8) == VLLM_OK) for (int i = 0; i < 8; ++i) out[4 + i] = pp[i];
        return 12;
    }
    return n < 8 ? n : 8;
}

}  // namespace vllm
