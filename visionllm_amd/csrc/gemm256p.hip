// bf16 GEMM, the 8-phase 256x256x64 schedule of gemm256.hip as a PERSISTENT kernel: one workgroup per CU walks over its
// tiles, and the refill pipeline never drains between them.
//
// Why (profiles/r02_gemm256_block_trace.txt, r02_gemm256_phases.txt; K = 1024, the ViT-L qkv / fc1 shapes): of the 30.7 us a
// CU spends per tile, 22.5 us are the main loop; the rest is the prologue (first K tiles from a cold ring, 1.9 us), the epilogue
// (output tile through LDS + waiting for the stores, 4.9 - 8.5 us) and the gap until the next workgroup starts (1.3 us).  Here
//   * the K-tile refills of the last two iterations of a tile fetch the first two K tiles of the block's NEXT tile (same ring,
//     same counted vmcnt(6) once per K tile): when the last MFMA of a tile retires, the next tile's first K tile is in LDS;
//   * the epilogue stores straight from the accumulators (after a lane-row exchange 16 bytes = 8 consecutive features per lane;
//     rows beyond M and the columns of a ragged last tile are dropped by the descriptor's range check) and does not wait for
//     the stores: they drain under the next tile's main loop.  vmcnt counts them together with the refills; loads return in
//     order among themselves, so "at most 6 outstanding" still means "every refill but the newest six has landed" -- the wait
//     can only be longer than needed, never shorter;
//   * operands are addressed through buffer descriptors: a lane's part of a refill address is one 32-bit offset per DMA slot
//     (4 registers in all), the tile / K-tile part is scalar -- nothing per-lane changes from tile to tile;
//   * K tile 0 of a tile takes the MFMA's constant-zero accumulator; the two staggered wave groups are level around the epilogue.
// The per-tile column vectors (bias, LayerScale; for the consumer of a folded norm the column sums, bias' and the rows'
// statistics) come in by LDS-DMA one tile ahead into double-buffered LDS behind the ring.  The residual epilogue reads the
// residual tile in the store layout with inline buffer loads and can produce a folded norm's row statistics (STATS).
//
// Scope: bias / GELU / quick-GELU / residual epilogues (qkv, fc1, proj, fc2 and the projector linears), N a multiple of 8, at
// least as many tiles as CUs, no CLS-skipping loader; the launcher in gemm256.hip asks for this schedule before it considers a
// stream-K tail.  Everything else stays on gemm256.hip.  VLLM_GEMM_PERSIST=0 / VLLM_GEMM_FORCE_TILEWISE switch it off.
// Measured and the four properties of the part / toolchain that the guards in this file are for: DESIGN.md section 3.2 and NOTES/rounds_1_to_4.md section 3.2 ("Four properties ...").
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

namespace {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int P_BN = 256, P_BK = 64, P_THREADS = 512;
constexpr int P_HALF = 128 * P_BK * 2;            // 16 KiB half-tile
constexpr int P_STAGE = 4 * P_HALF;               // A0 A1 B0 B1
constexpr int P_A0 = 0, P_A1 = P_HALF, P_B0 = 2 * P_HALF, P_B1 = 3 * P_HALF;
constexpr int P_RING = 2 * P_STAGE;               // 128 KiB
constexpr int P_TAB = P_RING;                     // folded norm: 256 x {r, -r mean}                        2 KiB
constexpr int P_COL = P_TAB + 2048;               // 2 x 2 KiB: bias (bf16) | folded norm: colsum[256], bias'[256] (fp32)
constexpr int P_RAW = P_COL + 2 * 2048;           // folded norm: 2 x 8 KiB of raw statistics (256 rows x 4 x {mean, M2})
constexpr int P_LDS = P_RAW + 2 * 8192;           // 150 KiB

__device__ unsigned long long g_p_prof[8];        // VLLM_GEMM_PROF: ticks of wave 0: main loops, epilogues, tiles, K tile 0 / 1 / 2 / 3 of a tile

// v_permlane16_swap_b32 a, b: the odd lane rows (16 lanes each) of a change places with the even lane rows of b --
//   a = [a.row0, b.row0, a.row2, b.row2],  b = [a.row1, b.row1, a.row3, b.row3].
// Inline, with its own wait states: (1) this compiler loses the SECOND result when the two are used separately (it reads both
// from the first register and reuses the second: seen in the disassembly of a four-line kernel), (2) a vector-memory read or a
// write of the registers right behind the instruction raced with it (last lanes of each lane row stale, run-to-run different).
__device__ __forceinline__ void lane_row_swap(unsigned &a, unsigned &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}

// v_permlane32_swap_b32 a, b: the upper 32 lanes of a change places with the lower 32 lanes of b (same guards as above).
__device__ __forceinline__ void lane_half_swap(unsigned &a, unsigned &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
// {mean, M2} of n values each, held by lane pairs 16 (STEP 16) or 32 (STEP 32) lanes apart -> {mean, M2} of the 2n values, in
// both lanes, combined in a fixed order (lower lanes first): Chan's update for equal counts.  rms: plain sums of squares.
template <int STEP>
__device__ __forceinline__ void pair_combine(float &mean, float &m2, float half_n, bool rms)
{
    unsigned ma = __builtin_bit_cast(unsigned, mean), mb = ma, qa = __builtin_bit_cast(unsigned, m2), qb = qa;
    if (STEP == 16) { lane_row_swap(ma, mb); lane_row_swap(qa, qb); } else { lane_half_swap(ma, mb); lane_half_swap(qa, qb); }
    const float fa = __builtin_bit_cast(float, ma), fb = __builtin_bit_cast(float, mb);
    const float ga = __builtin_bit_cast(float, qa), gb = __builtin_bit_cast(float, qb);
    if (rms) { mean = fa + fb; m2 = 0.f; return; }
    const float d = fb - fa;
    mean = fmaf(0.5f, d, fa);
    m2 = fmaf(d * d, half_n, ga + gb);
}

// One 16-byte output store.  The data registers stay untouched for 16 wait states behind it: with the next instruction but one
// writing the first of them (the compiler's hazard table has no entry for a 16-byte store with a scalar offset) the first
// dword came out stale in the last lanes of each lane row, run-to-run different, in the wave group that stores into a busy
// address path.
__device__ __forceinline__ void store_piece(u32x4_t &o, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)voff, (int)soff, 0);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(o));
}

#define P_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P_BARRIER()                       \
    do {                                  \
        __builtin_amdgcn_s_barrier();     \
        __builtin_amdgcn_sched_barrier(0);\
    } while (0)

// HALFT: the instantiation that can run the last round as half-height tiles (see the tile loop).  A separate instantiation because
// the second K loop costs the whole-tile loop registers (210 -> 237 VGPRs, 4 -> 33 spilled SGPRs) and 2 % of its speed (fc1, same box:
// 169 -> 173 us): launches whose last round stays whole keep the kernel they had.
template <int EPI, int MT, bool LNC, int STATS = 0, bool HALFT = false>   // STATS: 0 none, 1 {mean, M2} pairs per (row, tile), 2 wide (RMSNorm: one float per (row, tile), ragged last tile allowed)
__global__ __launch_bounds__(P_THREADS, 1) void gemm256p_kernel(const GemmArgs a)
{
    static_assert(EPI == EPI_BIAS || EPI == EPI_GELU || EPI == EPI_QUICK_GELU || (EPI == EPI_RESIDUAL && !LNC),
                  "persistent schedule: bias / GELU / quick-GELU / residual epilogues");
    constexpr bool RES = EPI == EPI_RESIDUAL;
    static_assert(!STATS || RES, "row statistics for a folded norm: the residual epilogue produces them");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, kq = lane >> 4;
    const int tid = wave * 64 + lane;
    const int nk = a.K / P_BK;
    const int T = a.mt * a.nt, G = gridDim.x;
    constexpr int BM = 64 * MT, HM = 32 * MT;     // block rows; rows of an A half

    // tile -> first row / column.  tile_rb == 0: the dense tile order of gemm256.hip (XCD = tile & 7 owns a panel).
    // tile_rb = RB > 0 (round 4): BANDED order.  The tiles are enumerated band by band (RB row panels), column by column inside a
    // band, and the 32 blocks of an XCD take 32 CONSECUTIVE tiles of that enumeration in every full round -- an RB x (32 / RB)
    // rectangle of the tile grid: its 32 CUs, which run in step, share RB activation panels and 32 / RB weight tiles through the
    // XCD's L2 (RB + 32 / RB operand streams per round instead of 1 + 32: in the dense order an XCD walks 32 different column
    // tiles of ONE row panel, and every XCD streams the whole weight once per row panel -- 13.7 GB through the fabric per
    // InternViT-6B fc1 launch against 1.39 GB of algorithmic bytes, profiles/pmc_traffic.json).  Same tiles, same arithmetic:
    // bit-identical outputs.  The last, partial round keeps the enumeration order (tile = block index).
    const int tile_rb = a.tile_rb;
    const int full_tiles = (T / G) * G;
    auto coords = [&](int td, int &m0, int &n0) {
        int tm, tn;
        if (tile_rb > 0) {
            int q = td;
            if (td < full_tiles) {
                const int cpx = G >> 3;                                  // blocks per XCD
                const int rnd = td / G, rem = td - rnd * G;
                q = ((rnd << 3) + (rem & 7)) * cpx + (rem >> 3);         // chunk (round, XCD) x position in the chunk
            }
            const int bs = tile_rb * a.nt;                                  // tiles per full band
            const int band = q / bs, r = q - band * bs;
            const int rows = min(tile_rb, a.mt - band * tile_rb);                 // (the last band may be shorter)
            tn = r / rows;
            tm = band * tile_rb + (r - tn * rows);
        } else if ((a.nt & 7) == 0) {
            const int xcd = td & 7, s = td >> 3, npx = a.nt >> 3;
            tn = xcd + 8 * (s % npx);
            tm = s / npx;
        } else {
            const int fullp = (a.mt >> 3) * 8 * a.nt;
            if (td < fullp) {
                const int xcd = td & 7, s = td >> 3;
                tm = xcd + 8 * (s / a.nt);
                tn = s % a.nt;
            } else {
                const int vx = a.mt & 7, e = td - fullp;
                tm = (a.mt & ~7) + e % vx;
                tn = e / vx;
            }
        }
        m0 = tm * BM; n0 = tn * P_BN;
    };

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, (int)(((unsigned)(a.M - 1) * (unsigned)a.ldx + (unsigned)a.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, (int)(((unsigned)(a.N - 1) * (unsigned)a.ldw + (unsigned)a.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.Y, 0, (int)(((unsigned)(a.M - 1) * (unsigned)a.ldy + (unsigned)a.N) * 2u), 0x00020000);

    // ---- a lane's part of the refill addresses: DMA slot s (0 / 1) of this wave is 8 rows of a half-tile; row r of the half,
    // 16-byte chunk c = (lane & 7) ^ (r & 7) of its 128 bytes (the source-side swizzle of gemm256.hip) ----
    unsigned xvo[2], wvo[2];
    int lslot[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int seg = wave * 2 + s;
        const int xs = seg < 4 * MT ? seg : 4 * MT - 1;           // (MT = 3: idle slots reload the last real segment)
        const int r8 = lane >> 3, c = (lane & 7) ^ (r8 & 7);
        xvo[s] = ((unsigned)(xs * 8 + r8) * (unsigned)a.ldx + (unsigned)c * 8u) * 2u;
        wvo[s] = ((unsigned)(seg * 8 + r8) * (unsigned)a.ldw + (unsigned)c * 8u) * 2u;
        lslot[s] = seg * 1024;
    }
    // refill of one half-tile: K tile `tt` of the current tile, or (tt >= nk) K tile tt - nk of the next one
    int m0 = 0, n0 = 0, m0n = 0, n0n = 0;
    auto issue_A = [&](int half, int stage, int tt) {
        const bool nx = tt >= nk;
        const unsigned so = ((unsigned)((nx ? m0n : m0) + half * HM) * (unsigned)a.ldx + (unsigned)((nx ? tt - nk : tt) * P_BK)) * 2u;
        char *dst = smem + stage * P_STAGE + (half ? P_A1 : P_A0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(dst + lslot[s]), 16, (int)xvo[s], (int)so, 0, 0);
    };
    auto issue_B = [&](int half, int stage, int tt) {
        const bool nx = tt >= nk;
        const unsigned so = ((unsigned)((nx ? n0n : n0) + half * 128) * (unsigned)a.ldw + (unsigned)((nx ? tt - nk : tt) * P_BK)) * 2u;
        char *dst = smem + stage * P_STAGE + (half ? P_B1 : P_B0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(dst + lslot[s]), 16, (int)wvo[s], (int)so, 0, 0);
    };
    // the column vectors (and, folded norm, the row statistics) of the tile at (mm, nn) into buffer `buf`: waves 0-1 / 0-7
    auto issue_vectors = [&](int mm, int nn, int buf) {
        // (the lane number is re-derived here, two VALU: as a value carried across the main loop it -- and the clamped column offset
        //  computed from it -- were what the register allocator spilled in the residual instantiations)
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        if constexpr (LNC) {
            if (!a.ln_wide) {   // wave w: rows 32 w .. 32 w + 31 of the tile, two lanes per row (16 of its 32 bytes each)
                int row = mm + wave * 32 + (lane >> 1);
                row = row < a.M ? row : a.M - 1;
                const float *src = a.ln_in + (size_t)row * 8 + (lane & 1) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + P_RAW + buf * 8192 + wave * 1024), 16, 0, 0);
            }
            if (wave < 2) {   // 256 floats = one 1 KiB instruction: wave 0 the column sums, wave 1 the biases (a missing vector stays zero)
                const float *vec = wave == 0 ? a.ln_colsum : a.ln_bias;
                int c4 = nn + lane * 4;
                c4 = c4 + 4 <= a.N ? c4 : a.N - 4;      // (ragged last column tile: in bounds, the lanes beyond N are not stored)
                if (vec) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vec + c4),
                                                          (__attribute__((address_space(3))) void *)(smem + P_COL + buf * 2048 + wave * 1024), 16, 0, 0);
            }
        } else {
            if (wave == 0 && a.bias) {   // 256 bf16 = 512 bytes: the upper 32 lanes re-read the last 16 bytes (harmless, in bounds)
                int c8 = nn + (lane < 32 ? lane : 31) * 8;
                c8 = c8 + 8 <= a.N ? c8 : a.N - 8;      // (ragged last column tile: in bounds, the lanes beyond N are not stored)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.bias + c8),
                                                 (__attribute__((address_space(3))) void *)(smem + P_COL + buf * 2048), 16, 0, 0);
            }
            if (RES && wave == 1 && a.scale) {   // LayerScale of the residual epilogue, the same way, 1 KiB further
                int c8 = nn + (lane < 32 ? lane : 31) * 8;
                c8 = c8 + 8 <= a.N ? c8 : a.N - 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.scale + c8),
                                                 (__attribute__((address_space(3))) void *)(smem + P_COL + buf * 2048 + 1024), 16, 0, 0);
            }
        }
    };
    // Wide statistics (round 5; GemmArgs::ln_wide: RMSNorm rows of up to 16 column tiles, 16 floats = 64 bytes per row): the 256 rows
    // of a tile are 16 KiB -- the whole raw region, ONE buffer: a tile's statistics are requested only after the previous tile's row
    // table has been built from it (they are needed a whole main loop later).  Wave w: rows 32 w .. 32 w + 31, four lanes per row,
    // two instructions.
    auto issue_stats_wide = [&](int mm) {
        if constexpr (LNC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int row = mm + wave * 32 + i * 16 + (lane >> 2);
                row = row < a.M ? row : a.M - 1;
                const float *src = a.ln_in + (size_t)row * 16 + (lane & 3) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + P_RAW + (wave * 2 + i) * 1024), 16, 0, 0);
            }
        }
    };

    // per-lane LDS byte offsets of the fragments inside a half-tile (ks = 0 / 1 differ by XOR 4 chunks)
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
    // a lane's part of the output address: row wr * 16 MT + fr of a quadrant piece; after the lane-row exchange of the epilogue
    // lane row kq holds the 8 features from column {0, 16, 8, 24}[kq] of the wave's 32
    const int ycol = wc * 32 + (kq & 1) * 16 + (kq >> 1) * 8;
    const unsigned yvo = ((unsigned)(wr * (16 * MT) + fr) * (unsigned)a.ldy + (unsigned)ycol) * 2u;

    // residual epilogue: the residual tile is read in the SAME lane layout the stores use (16 bytes = 8 features of one row per
    // lane), with inline buffer loads: a compiler-visible load in front of the stores would make every use wait for vmcnt(0)
    // (loads and stores retire out of order with each other), i.e. for the store just issued.  Descriptor built by hand for the asm.
    const u32x4_t rrs = {(unsigned)(reinterpret_cast<uintptr_t>(a.res) & 0xffffffffu), (unsigned)(reinterpret_cast<uintptr_t>(a.res) >> 32) & 0xffffu,
                         RES ? ((unsigned)(a.M - 1) * (unsigned)a.ldr + (unsigned)a.N) * 2u : 0u, 0x00020000u};
    const unsigned rvo = ((unsigned)(wr * (16 * MT) + fr) * (unsigned)a.ldr + (unsigned)ycol) * 2u;
    u32x4_t ra[RES ? MT : 1], rb[RES ? 3 * MT : 1];   // residual pieces of quadrant 0 / of quadrants 1, 2, 3
    auto res_load_q = [&](u32x4_t *dst, int q, int mm, int nn) {
        if constexpr (RES) {
            const int qi = q >> 1, qj = q & 1;
            const unsigned vo = nn + qj * 128 + ycol + 8 <= a.N ? rvo : 0x80000000u;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const unsigned so = ((unsigned)(mm + qi * HM + j * 16) * (unsigned)a.ldr + (unsigned)(nn + qj * 128)) * 2u;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst[j]) : "v"(vo), "s"(rrs), "s"(so) : "memory");
            }
        }
    };

    bf16x8_t wg[2][2], xf[MT][2], wf[2][2];
    f32x4_t acc[4][2][MT];   // [quadrant q = 2 * (A half) + (B half)][n tile][m tile]

    auto read_x = [&](const char *half) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            xf[t][0] = *reinterpret_cast<const bf16x8_t *>(half + xoff[t]);
            xf[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (xoff[t] ^ 64));
        }
    };
    auto read_w_into = [&](const char *half, bf16x8_t (&dst)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dst[t][0] = *reinterpret_cast<const bf16x8_t *>(half + woff[t]);
            dst[t][1] = *reinterpret_cast<const bf16x8_t *>(half + (woff[t] ^ 64));
        }
    };
    // (FIRST: K tile 0 of a tile starts its accumulators from the instruction's constant zero -- no 128 moves per tile)
#define P_MMA(Q, WF, FIRST)                                                                                     \
    do {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                  \
                    acc[Q][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i][ks], xf[j][ks],                \
                                                                           (FIRST && ks == 0) ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[Q][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    // ---- the block's tiles: iteration `it` is tile blockIdx.x + it * G in the whole rounds.  The LAST, incomplete round (r = T % G
    // tiles) has two forms: one tile per block for the first r blocks (whole tiles), or -- half_tail (round 4; the launcher sets it
    // when 2 r <= G) -- HALF tiles: block b < 2 r takes rows [h HM, (h + 1) HM) of tile full_tiles + b % r, h = b / r: the round then
    // costs a half tile's time (K tiles of two quadrants instead of four, three refilled half-tiles instead of four) instead of a
    // whole one while most of the CUs idle (qkv at 40 tiles: 4.27 rounds of work in 5).  Same arithmetic per output element.
    const int n_whole = T / G, r_last = T - n_whole * G;
    // (not in the residual instantiations: their 256-row form sits at the register limit -- with the second K loop it spills, and a
    //  spilled register that is the destination of a residual load still in flight is stored before the load lands: run-to-run
    //  different results, seen on hardware.  Their launches keep whole tiles.)
    constexpr bool HT = HALFT && !RES;
    const bool half_tail = HT && a.half_tail != 0 && r_last > 0;
    const int n_it = n_whole + ((int)blockIdx.x < (half_tail ? 2 * r_last : r_last) ? 1 : 0);
    auto locate = [&](int it, int &mm, int &nn) -> bool {   // -> is iteration `it` a half tile?
        if (it < n_whole || !half_tail) { coords((int)blockIdx.x + it * G, mm, nn); return false; }
        const int h = (int)blockIdx.x / r_last;
        coords(full_tiles + ((int)blockIdx.x - h * r_last), mm, nn);
        mm += h * HM;
        return true;
    };
    // ---- block prologue: the vectors of the first tile, its K tile 0 complete in stage 0, K tile 1's A0, B1, A1 in flight ----
    int it_cur = 0;
    bool half_cur = locate(0, m0, n0), half_nxt = false;
    if (n_it > 1) half_nxt = locate(1, m0n, n0n); else { m0n = m0; n0n = n0; }
    for (int i = tid; i < (P_LDS - P_RING) / 4; i += P_THREADS) reinterpret_cast<unsigned *>(smem + P_RING)[i] = 0u;   // (missing vectors read as zeros)
    __syncthreads();
    issue_vectors(m0, n0, 0);
    if (LNC && a.ln_wide) issue_stats_wide(m0);
    issue_A(0, 0, 0); issue_B(0, 0, 0); issue_B(1, 0, 0); issue_A(1, 0, 0);
    issue_A(0, 1, 1); issue_B(1, 1, 1); issue_A(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    P_BARRIER();

    unsigned pf[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int par = 0;                // stage of the current tile's K tile 0 (K tiles alternate stages across tile boundaries)
    for (int it = 0;; ++it) {
        const unsigned t_a = a.prof ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
        unsigned t_k[4] = {t_a, t_a, t_a, t_a};
        // Inside a tile group 1 runs one barrier behind group 0 (gemm256.hip: one group's MFMA section over the other's reads);
        // around the epilogue the groups are level again.  Left staggered, group 1 sits at a barrier through group 0's epilogue
        // and group 0 through group 1's: the two epilogues ran one after the other (phase clock: the first two K tiles of a
        // tile took 16.8 K ticks instead of 5.2 K).
        if (wr == 1) P_BARRIER();
        auto k_tile = [&](int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int s = (t + par) & 1;
            const char *st = smem + s * P_STAGE;
            // phase 1: quadrant (A0,B0); refill B0 of the OTHER stage with K tile t+1
            read_x(st + P_A0); read_w_into(st + P_B0, wf);
            issue_B(0, s ^ 1, t + 1);
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(0, wf, FIRST);
            P_BARRIER();
            // phase 2: quadrant (A0,B1); refill A0 (this stage) with K tile t+2
            read_w_into(st + P_B1, wg);
            issue_A(0, s, t + 2);
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(1, wg, FIRST);
            P_BARRIER();
            // phase 3: quadrant (A1,B1); refill B1 with K tile t+2
            read_x(st + P_A1);
            issue_B(1, s, t + 2);
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(3, wg, FIRST);
            P_BARRIER();
            // phase 4: quadrant (A1,B0); refill A1 with K tile t+2; retire everything but the last 3 half-tiles
            issue_A(1, s, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(2, wf, FIRST);
            P_BARRIER();
            if (a.prof && t < 4) t_k[t] = (unsigned)__builtin_amdgcn_s_memtime();   // (clock: the first K tiles of a tile, one by one)
        };
        // a HALF tile's K tile: quadrants (A0,B0), (A0,B1) only.  Phases 1 and 2 are the full tile's (same refills: B0 of the other stage
        // with K tile t + 1, A0 of this stage with t + 2); the third phase has no MFMA section: it refills B1 (read for the last time in
        // phase 2, two barriers ago: the full tile's distance) and retires every refill but the newest two half-tiles -- B0 of K tile
        // t + 1 has landed when the next K tile starts, as behind the full tile's phase 4.  A1 is never read and never refilled.
        auto k_tile_half = [&](int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int s = (t + par) & 1;
            const char *st = smem + s * P_STAGE;
            read_x(st + P_A0); read_w_into(st + P_B0, wf);
            issue_B(0, s ^ 1, t + 1);
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(0, wf, FIRST);
            P_BARRIER();
            read_w_into(st + P_B1, wg);
            issue_A(0, s, t + 2);
            P_WAIT_LGKM0(); P_BARRIER();
            P_MMA(1, wg, FIRST);
            P_BARRIER();
            issue_B(1, s, t + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            P_WAIT_LGKM0(); P_BARRIER();
            P_BARRIER();
            if (a.prof && t < 4) t_k[t] = (unsigned)__builtin_amdgcn_s_memtime();
        };
        if (HT && half_cur) {
            // (no early request of the residual pieces here: registers in flight must not be DEFINED on both sides of a join -- where
            //  the allocator picks different registers it copies them while the load is still out; a half tile asks for its pieces at
            //  the start of its epilogue, inside one block with their wait)
            k_tile_half(0, std::true_type{});
#pragma unroll 1
            for (int t = 1; t < nk; ++t) k_tile_half(t, std::false_type{});
            // (quadrants 2, 3 are not computed: DEFINE them, or their registers stay live around the whole tile loop -- +64 VGPRs, spills)
#pragma unroll
            for (int q = 2; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[q][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        } else {
        k_tile(0, std::true_type{});
        if constexpr (RES) {
            // the residual pieces of quadrant 0 are requested in front of the last K tile (more than one quadrant's worth does
            // not fit beside the fragments: the compiler would spill the in-flight registers).  That K tile's counted wait is
            // exact for the refills only (nothing is assumed about the order in which register loads and LDS-DMA retire relative
            // to each other): the epilogue waits for the pieces with vmcnt(0).
#pragma unroll 1
            for (int t = 1; t < nk - 1; ++t) k_tile(t, std::false_type{});
            res_load_q(ra, 0, m0, n0);
            k_tile(nk - 1, std::false_type{});
        } else {
#pragma unroll 1
            for (int t = 1; t < nk; ++t) k_tile(t, std::false_type{});
        }
        }
        if (wr == 0) P_BARRIER();   // level again
        par = (par + nk) & 1;
        const unsigned t_b = a.prof ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;

        // ---- between two tiles: the ring already holds the next tile's K tile 0 (its K tile 1 is in flight) ----
        const bool more = it_cur + 1 < n_it;
        const int buf = it & 1;
        const int nq = (HT && half_cur) ? 2 : 4;   // quadrants this tile stores (a half tile: the A0 row half only)
        if (more) issue_vectors(m0n, n0n, buf ^ 1);   // (that buffer was last read in the epilogue before the main loop just finished)
        if constexpr (LNC) {
            // the row table {r, -r mean} of this tile from its statistics (Chan's update in a fixed order, constants from the
            // launcher)
            if (a.ln_wide) {
                if (tid < BM) {   // 16 slots of row tid (unused ones are zero: cleared once per call by the orchestrator), a fixed tree
                    f32x4_t w0, w1, w2, w3;
                    const unsigned addr = (unsigned)(size_t)(smem + P_RAW + tid * 64);
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(addr) : "memory");
                    const float ss = (((w0[0] + w0[1]) + (w0[2] + w0[3])) + ((w1[0] + w1[1]) + (w1[2] + w1[3]))) +
                                     (((w2[0] + w2[1]) + (w2[2] + w2[3])) + ((w3[0] + w3[1]) + (w3[2] + w3[3])));
                    reinterpret_cast<float2_t *>(smem + P_TAB)[tid] = (float2_t){__builtin_amdgcn_rsqf(fmaf(ss, a.ln_inv_cols, a.ln_eps)), 0.f};
                }
                P_WAIT_LGKM0(); P_BARRIER();
                if (more) issue_stats_wide(m0n);   // (the raw region is free again: every reader is behind the barrier)
            } else {
            if (tid < BM) {
                // (read with inline ds_read: for a compiler-visible LDS load behind an LDS-DMA the compiler drains vmcnt to 0 --
                //  here that would wait for the next tile's refills; the statistics landed a whole main loop ago)
                float2_t sp[4];
                {
                    const unsigned addr = (unsigned)(size_t)(smem + P_RAW + buf * 8192 + tid * 32);
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(sp[0]), "=&v"(sp[1]), "=&v"(sp[2]), "=&v"(sp[3]) : "v"(addr) : "memory");
                }
                float r_, nrm_;
                if (a.ln_rms) {
                    const float ss = (sp[0].x + sp[1].x) + (sp[2].x + sp[3].x);
                    r_ = __builtin_amdgcn_rsqf(fmaf(ss, a.ln_inv_cols, a.ln_eps));
                    nrm_ = 0.f;
                } else {
                    float mean = sp[0].x, m2 = sp[0].y;
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
                reinterpret_cast<float2_t *>(smem + P_TAB)[tid] = (float2_t){r_, nrm_};
            }
            P_WAIT_LGKM0(); P_BARRIER();
            }
        }
        if constexpr (RES) {
            // ---- residual epilogue: y = res + (acc + bias) * LayerScale, one rounding.  The lane-row exchange is done on the fp32
            // values (four exchanges per piece), the residual added in the exchanged layout.  Quadrant 0's residual pieces were
            // requested a K tile ago (waited for here); those of quadrants 1-3 are requested now (the fragment registers are
            // free), in front of every store of this tile, and waited for once (vmcnt(0)) behind quadrant 0.
            if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]) :: "memory");
            res_load_q(rb, 1, m0, n0); res_load_q(rb + MT, 2, m0, n0); res_load_q(rb + 2 * MT, 3, m0, n0);
            const char *colb = smem + P_COL + buf * 2048;
            const bool scaled = a.scale != nullptr;
            u32x4_t o_prev = {0u, 0u, 0u, 0u};
            unsigned so_prev = 0u, yv_prev = 0u;
            unsigned yv[2];
            const bool rms_ = STATS == 2 ? true : (a.ln_rms != 0);   // (wide statistics exist for RMSNorm only: a constant there)
            float st_m[STATS ? MT : 1], st_q[STATS ? MT : 1];   // a lane's {mean, M2} of the qj = 0 pieces, until their qj = 1 partners
#pragma unroll
            for (int qj = 0; qj < 2; ++qj) yv[qj] = n0 + qj * 128 + ycol + 8 <= a.N ? yvo : 0x80000000u;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < 2 || nq == 4) {   // (q < 2: compile-time true -- no join inside the in-flight windows above)
                const int qi = q >> 1, qj = q & 1;
                if (q == 1) {
                    if constexpr (MT == 4)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]),
                                                            "+v"(rb[8]), "+v"(rb[9]), "+v"(rb[10]), "+v"(rb[11]) :: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]),
                                                            "+v"(rb[8]) :: "memory");
                }
                float bia[2][4], scl[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int nl = qj * 128 + wc * 32 + i * 16 + kq * 4;
                    const uint2_t b = *reinterpret_cast<const uint2_t *>(colb + nl * 2);
                    bia[i][0] = bf16lo_to_f32(b.x); bia[i][1] = bf16hi_to_f32(b.x); bia[i][2] = bf16lo_to_f32(b.y); bia[i][3] = bf16hi_to_f32(b.y);
                    const uint2_t sc = *reinterpret_cast<const uint2_t *>(colb + 1024 + nl * 2);
                    scl[i][0] = bf16lo_to_f32(sc.x); scl[i][1] = bf16hi_to_f32(sc.x); scl[i][2] = bf16lo_to_f32(sc.y); scl[i][3] = bf16hi_to_f32(sc.y);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    float y[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v0 = acc[q][0][j][k] + bia[0][k], v1 = acc[q][1][j][k] + bia[1][k];
                        if (scaled) { v0 *= scl[0][k]; v1 *= scl[1][k]; }
                        unsigned s0 = __builtin_bit_cast(unsigned, v0), s1 = __builtin_bit_cast(unsigned, v1);
                        lane_row_swap(s0, s1);
                        y[k] = __builtin_bit_cast(float, s0); y[4 + k] = __builtin_bit_cast(float, s1);
                    }
                    const u32x4_t rr = q == 0 ? ra[j] : rb[(q - 1) * MT + j];
                    y[0] += bf16lo_to_f32(rr.x); y[1] += bf16hi_to_f32(rr.x); y[2] += bf16lo_to_f32(rr.y); y[3] += bf16hi_to_f32(rr.y);
                    y[4] += bf16lo_to_f32(rr.z); y[5] += bf16hi_to_f32(rr.z); y[6] += bf16lo_to_f32(rr.w); y[7] += bf16hi_to_f32(rr.w);
                    if (q + j > 0 && !(a.prof & 2)) store_piece(o_prev, yrs, yv_prev, so_prev);
                    o_prev = (u32x4_t){pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
                    if constexpr (STATS) {
                        // folded norm, producer side: {mean, M2} ({sum of squares, -} for RMSNorm) of the bf16 values just packed,
                        // per output row and 256-column tile -- 8 values here, the row's other 8 of this lane when qj = 1, then
                        // the 4 lane rows, then (through LDS) the 4 waves that share the row
                        const unsigned u[4] = {o_prev.x, o_prev.y, o_prev.z, o_prev.w};
                        float x_[8];
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) { x_[2 * k2] = bf16lo_to_f32(u[k2]); x_[2 * k2 + 1] = bf16hi_to_f32(u[k2]); }
                        float pm, pq = 0.f;
                        if (rms_) {
                            pm = 0.f;
#pragma unroll
                            for (int k2 = 0; k2 < 8; ++k2) pm = fmaf(x_[k2], x_[k2], pm);
                            if constexpr (STATS == 2) { if (yv[qj] == 0x80000000u) pm = 0.f; }   // (ragged last column tile: columns beyond N do not exist)
                        } else {
                            pm = ((x_[0] + x_[1]) + (x_[2] + x_[3])) + ((x_[4] + x_[5]) + (x_[6] + x_[7]));
                            pm *= 0.125f;
#pragma unroll
                            for (int k2 = 0; k2 < 8; ++k2) { const float d = x_[k2] - pm; pq = fmaf(d, d, pq); }
                        }
                        if (qj == 0) { st_m[j] = pm; st_q[j] = pq; }
                        else {
                            float mean = st_m[j], m2 = st_q[j];
                            if (rms_) mean += pm;
                            else { const float d = pm - mean; mean = fmaf(0.5f, d, mean); m2 = fmaf(d * d, 4.f, m2 + pq); }
                            pair_combine<16>(mean, m2, 8.f, rms_);
                            pair_combine<32>(mean, m2, 16.f, rms_);
                            if (kq == 0) reinterpret_cast<float2_t *>(smem + P_RAW)[(qi * HM + wr * (16 * MT) + j * 16 + fr) * 4 + wc] = (float2_t){mean, m2};
                        }
                    }
                    yv_prev = yv[qj];
                    so_prev = ((unsigned)(m0 + qi * HM + j * 16) * (unsigned)a.ldy + (unsigned)(n0 + qj * 128)) * 2u;
                }
            }
            if (!(a.prof & 2)) store_piece(o_prev, yrs, yv_prev, so_prev);
            if constexpr (STATS) {
                P_WAIT_LGKM0(); P_BARRIER();
                if (tid < BM) {   // the four waves' partials (64 values each) of row tid, in a fixed order
                    float2_t pp[4];
                    {
                        const unsigned addr = (unsigned)(size_t)(smem + P_RAW + tid * 32);   // (inline: see the folded-norm consumer)
                        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(pp[0]), "=&v"(pp[1]), "=&v"(pp[2]), "=&v"(pp[3]) : "v"(addr) : "memory");
                    }
                    float2_t r_;
                    if (rms_) r_ = (float2_t){(pp[0].x + pp[1].x) + (pp[2].x + pp[3].x), 0.f};
                    else {
                        const float d0 = pp[1].x - pp[0].x, d1 = pp[3].x - pp[2].x;
                        const float m0_ = fmaf(0.5f, d0, pp[0].x), m1_ = fmaf(0.5f, d1, pp[2].x);
                        const float q0_ = fmaf(d0 * d0, 32.f, pp[0].y + pp[1].y), q1_ = fmaf(d1 * d1, 32.f, pp[2].y + pp[3].y);
                        const float d = m1_ - m0_;
                        r_ = (float2_t){fmaf(0.5f, d, m0_), fmaf(d * d, 64.f, q0_ + q1_)};
                    }
                    const int m = m0 + tid;
                    if constexpr (STATS == 2) {   // (slots >= nt of a row are never written: the orchestrator clears the buffer once per call)
                        if (m < a.M) a.ln_out[(size_t)m * 16 + (n0 >> 8)] = r_.x;
                    } else if (m < a.M) *reinterpret_cast<float2_t *>(a.ln_out + ((size_t)m * a.nt + (n0 >> 8)) * 2) = r_;
                }
            }
        } else
        // ---- epilogue: straight from the accumulators, nothing waits for the stores ----
        // A lane holds 4 features (8 bytes packed) of row fr in each of the wave's two 16-column n tiles; v_permlane16_swap
        // exchanges the odd lane rows of the first with the even lane rows of the second, after which a lane owns 8 CONSECUTIVE
        // features: one 16-byte store per lane, 64 contiguous bytes per output row and instruction (half the instructions and
        // twice the segment length of the plain accumulator layout -- the stores share the CU's address path with the refills).
        {
            const char *colb = smem + P_COL + buf * 2048;
            float2_t rn[LNC ? 2 : 1][LNC ? MT : 1];
            if constexpr (LNC) {
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                    for (int j = 0; j < MT; ++j) rn[qi][j] = reinterpret_cast<const float2_t *>(smem + P_TAB)[qi * HM + wr * (16 * MT) + j * 16 + fr];
            }
            u32x4_t o_prev = {0u, 0u, 0u, 0u};
            unsigned so_prev = 0u, yv_prev = 0u;
            // last column tile of an N that is not a multiple of 256: a lane whose 8 features lie beyond N stores out of the
            // descriptor's range (dropped)
            unsigned yv[2];
#pragma unroll
            for (int qj = 0; qj < 2; ++qj) yv[qj] = n0 + qj * 128 + ycol + 8 <= a.N ? yvo : 0x80000000u;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < 2 || nq == 4) {   // (q < 2: compile-time true -- no join inside the in-flight windows above)
                const int qi = q >> 1, qj = q & 1;
                EpiCols cols[2];
                f32x4_t csum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int nl = qj * 128 + wc * 32 + i * 16 + kq * 4;
                    if constexpr (LNC) {
                        csum[i] = *reinterpret_cast<const f32x4_t *>(colb + nl * 4);
                        const f32x4_t b4 = *reinterpret_cast<const f32x4_t *>(colb + 1024 + nl * 4);
                        cols[i].bia[0] = b4[0]; cols[i].bia[1] = b4[1]; cols[i].bia[2] = b4[2]; cols[i].bia[3] = b4[3];
                    } else {
                        const uint2_t b = *reinterpret_cast<const uint2_t *>(colb + nl * 2);
                        cols[i].bia[0] = bf16lo_to_f32(b.x); cols[i].bia[1] = bf16hi_to_f32(b.x); cols[i].bia[2] = bf16lo_to_f32(b.y); cols[i].bia[3] = bf16hi_to_f32(b.y);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) cols[i].scl[r] = 1.f;
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[4];
                        if constexpr (LNC) epi_value_folded<EPI>(acc[q][i][j], rn[qi][j].x, rn[qi][j].y, csum[i], cols[i], v);
                        else epi_value<EPI>(a, 0, 0, acc[q][i][j], cols[i], v);
                        pk[i][0] = pack_bf16x2(v[0], v[1]); pk[i][1] = pack_bf16x2(v[2], v[3]);
                    }
                    lane_row_swap(pk[0][0], pk[1][0]);
                    lane_row_swap(pk[0][1], pk[1][1]);
                    // the store of a piece goes out one piece LATER (behind the next piece's arithmetic): issued right behind
                    // the lane-row exchange it picked up stale data in the last lanes of every lane row (rows fr >= 12 wrong,
                    // run-to-run different)
                    if (q + j > 0 && !(a.prof & 2)) store_piece(o_prev, yrs, yv_prev, so_prev);   // (prof bit 1: ablation, no stores)
                    o_prev = (u32x4_t){pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                    yv_prev = yv[qj];
                    so_prev = ((unsigned)(m0 + qi * HM + j * 16) * (unsigned)a.ldy + (unsigned)(n0 + qj * 128)) * 2u;
                }
            }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (the same distance for the last piece)
            if (!(a.prof & 2)) store_piece(o_prev, yrs, yv_prev, so_prev);
        }
        if (a.prof) {   // (summed per block, written once at its end: an atomic per tile would sit in front of the next tile's refills)
            const unsigned t_c = (unsigned)__builtin_amdgcn_s_memtime();
            pf[0] += t_b - t_a; pf[1] += t_c - t_b; pf[2] += 1u;
            pf[3] += t_k[0] - t_a; pf[4] += t_k[1] - t_k[0]; pf[5] += t_k[2] - t_k[1]; pf[6] += t_k[3] - t_k[2];
        }
        if (!more) break;
        ++it_cur; m0 = m0n; n0 = n0n; half_cur = half_nxt;
        if (it_cur + 1 < n_it) half_nxt = locate(it_cur + 1, m0n, n0n); else { m0n = m0; n0n = n0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
    if (a.prof && wave == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&g_p_prof[i], (unsigned long long)pf[i]);
    }
}

bool persist_disabled()
{
    static const int v = [] { const char *e = getenv("VLLM_GEMM_PERSIST"); return e && e[0] == '0' ? 1 : 0; }();
    return v != 0;
}

}  // namespace

// Whether the persistent schedule takes this GEMM (the 8-phase launcher asks BEFORE it considers a stream-K tail, with the tile
// height that needs fewer whole rounds).
bool gemm256p_takes(int epi, const GemmArgs &a, int cus)
{
    if (persist_disabled() || a.no_persist || (cus & 7) != 0) return false;
    if (!(epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_QUICK_GELU || epi == EPI_RESIDUAL)) return false;
    if (a.xP != 0 || a.sk_tiles > 0 || a.variant256 == 5) return false;
    if (a.ln_out && (epi != EPI_RESIDUAL || ((a.N % P_BN) != 0 && !a.ln_wide) || (reinterpret_cast<uintptr_t>(a.ln_out) & 7u) != 0)) return false;
    if (a.ln_wide && ((a.ln_in && (!a.ln_rms || a.ln_slots > 16 || (reinterpret_cast<uintptr_t>(a.ln_in) & 15u) != 0)) ||
                      (a.ln_out && (!a.ln_rms || (a.N + P_BN - 1) / P_BN > 16 || (a.N & 7) != 0)))) return false;
    if ((a.N & 7) != 0 || a.N < P_BN || (a.K % P_BK) != 0 || a.K < 2 * P_BK || (a.ldy & 3) != 0 || (a.ldx & 7) != 0 || (a.ldw & 7) != 0) return false;
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!al16(a.X) || !al16(a.W) || !al16(a.Y) || (a.bias && !al16(a.bias))) return false;
    if (epi == EPI_RESIDUAL && (a.ln_in || !a.res || !al16(a.res) || (a.ldr & 3) != 0 || (a.scale && !al16(a.scale)) ||
                                ((unsigned long long)a.M + 256) * (unsigned long long)a.ldr * 2 >= (1ull << 31))) return false;
    const unsigned long long lim = 1ull << 31;   // byte offsets are 32-bit and must stay clear of the range check's wrap
    if (((unsigned long long)a.M + 256) * (unsigned long long)a.ldx * 2 >= lim || ((unsigned long long)a.N + 256) * (unsigned long long)a.ldw * 2 >= lim ||
        ((unsigned long long)a.M + 256) * (unsigned long long)a.ldy * 2 >= lim) return false;
    if ((long)a.mt * a.nt < cus) return false;   // fewer tiles than CUs: nothing to pipeline across
    return true;
}

static long g_p_half_launches = 0;   // launches whose last round ran as half-height tiles (tests assert the path they mean to cover ran)
long gemm256p_half_launches() { return g_p_half_launches; }
static int g_half_tail = -1;
int gemm_half_tail()
{
    if (g_half_tail < 0) {
        const char *e = getenv("VLLM_GEMM_HALF_TAIL");
        g_half_tail = e ? atoi(e) != 0 : 1;
    }
    return g_half_tail;
}
int gemm_half_tail_set(int v) { const int old = gemm_half_tail(); g_half_tail = v != 0; return old; }
static long g_p_launches = 0;   // (vllm_gemm_persistent_launches: tests assert the path they mean to cover ran)
long gemm256p_launches() { return g_p_launches; }

int gemm256p_launch(int epi, int MT, const GemmArgs &a_, int cus, hipStream_t st)
{
    GemmArgs a = a_;
    {
        // automatic (measured, tools/gemm_tile_order_ab.py, profiles/r04_gemm_tile_order.txt): bands of 4 row panels (4 x 8 tiles
        // per XCD and round) from 32 column tiles on, bands of 8 (8 x 4) from 8 column tiles on, the dense order below that
        // (N = 1024: 4 column tiles -- nothing to share)
        const int e = gemm_tile_rb();   // -1: automatic
        a.tile_rb = e >= 0 ? e : (a.nt >= 32 ? 4 : a.nt >= 8 ? 8 : 0);
        if (a.tile_rb > 32 || (a.tile_rb > 0 && (cus & 7) != 0)) a.tile_rb = 0;
    }
    ++g_p_launches;
    const long T = (long)a.mt * a.nt;
    {   // half-height tiles in the last, incomplete round (kernel comment): when its tiles, cut in two, still fit the grid
        const long r = T % cus;
        a.half_tail = (gemm_half_tail() && epi != EPI_RESIDUAL && T >= cus && r > 0 && 2 * r <= cus) ? 1 : 0;
        if (a.half_tail) ++g_p_half_launches;
    }
    const dim3 grid((unsigned)std::min<long>(T, cus)), block(P_THREADS);
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
#define SETATTR(E, L) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<E, 4, L>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS); \
                      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<E, 3, L>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS); \
                      if (E != EPI_RESIDUAL) { \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<E == EPI_RESIDUAL ? EPI_BIAS : E, 4, L, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS); \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<E == EPI_RESIDUAL ? EPI_BIAS : E, 3, L, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS); }
        SETATTR(EPI_BIAS, false); SETATTR(EPI_GELU, false); SETATTR(EPI_QUICK_GELU, false);
        SETATTR(EPI_BIAS, true); SETATTR(EPI_GELU, true); SETATTR(EPI_QUICK_GELU, true); SETATTR(EPI_RESIDUAL, false);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<EPI_RESIDUAL, 4, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<EPI_RESIDUAL, 3, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<EPI_RESIDUAL, 4, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm256p_kernel<EPI_RESIDUAL, 3, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
#undef SETATTR
    }
#define LAUNCH_H(E, H) do { if (a.ln_in) { if (MT == 4) VLLM_LAUNCH((gemm256p_kernel<E, 4, true, 0, H>), grid, block, P_LDS, st, a); \
                                            else VLLM_LAUNCH((gemm256p_kernel<E, 3, true, 0, H>), grid, block, P_LDS, st, a); } \
                            else { if (MT == 4) VLLM_LAUNCH((gemm256p_kernel<E, 4, false, 0, H>), grid, block, P_LDS, st, a); \
                                   else VLLM_LAUNCH((gemm256p_kernel<E, 3, false, 0, H>), grid, block, P_LDS, st, a); } } while (0)
#define LAUNCH(E) do { if (a.half_tail) LAUNCH_H(E, true); else LAUNCH_H(E, false); } while (0)
    if (epi == EPI_RESIDUAL && a.ln_out && a.ln_wide) {
        if (MT == 4) VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 4, false, 2>), grid, block, P_LDS, st, a);
        else VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 3, false, 2>), grid, block, P_LDS, st, a);
    } else if (epi == EPI_RESIDUAL && a.ln_out) {
        if (MT == 4) VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 4, false, 1>), grid, block, P_LDS, st, a);
        else VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 3, false, 1>), grid, block, P_LDS, st, a);
    } else if (epi == EPI_RESIDUAL) {
        if (MT == 4) VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 4, false>), grid, block, P_LDS, st, a);
        else VLLM_LAUNCH((gemm256p_kernel<EPI_RESIDUAL, 3, false>), grid, block, P_LDS, st, a);
    } else if (epi == EPI_BIAS) LAUNCH(EPI_BIAS); else if (epi == EPI_GELU) LAUNCH(EPI_GELU); else LAUNCH(EPI_QUICK_GELU);
#undef LAUNCH
#undef LAUNCH_H
    VLLM_CHECK_LAUNCH("gemm256p_kernel");
    return VLLM_OK;
}

int gemm256p_debug_counters(long *out, int n)
{
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_p_prof), sizeof(h)) != hipSuccess) return VLLM_ELAUNCH;
    for (int i = 0; i < n && i < 8; ++i) out[i] = (long)h[i];
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p_prof), z, sizeof(z));
    return VLLM_OK;
}

}  // namespace vllm
