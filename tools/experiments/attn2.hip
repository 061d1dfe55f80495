// Fused multi-head self-attention forward, schedule 2 (round 3): the tiles, fragment layouts and arithmetic of attn.hip
// (transposed scores S^T = K Q^T on v_mfma_f32_32x32x16_bf16, in-register online softmax in the exp2 domain with deferred
// rescale, V read with ds_read_b64_tr_b16, O^T += V^T P^T), with the instruction schedule taken out of the compiler's hands.
//
// Replaces  FlashAttention.forward / flash_attn_varlen_qkvpacked_func
//             (VisionLLMv2/visionllmv2/model/internvit/flash_attention.py:30-75; causal=False, dropout 0,
//              softmax_scale = d^-0.5) and InternAttention._naive_attn (modeling_intern_vit.py:126-143).
//
// What the ISA of attn_fwd_kernel showed (round 3): hipcc serialises the QK^T product as
// {ds_read_b128 -> s_waitcnt lgkmcnt(0) -> v_mfma} x 16 through ONE fragment register (an LDS round trip per MFMA), and
// puts an s_waitcnt vmcnt(0) in front of the first V transpose-read -- in the middle of the softmax -- which drains the
// LDS-DMA of the NEXT tile issued a few hundred cycles earlier (the LDS read "may alias" the DMA in flight).  Here
//   * every LDS fragment read is inline asm with its own counted s_waitcnt (the compiler neither tracks nor "protects"
//     them); results of MFMA clusters are pinned with empty asm statements, because a pure instruction is otherwise placed
//     just above its first user, whatever fences lie in between;
//   * a KV tile is:  barrier | DMA(t+1) | K fragments -> registers in ONE batch | QK^T: 2 KS MFMAs back to back, the two
//     key blocks' accumulators alternating | V fragments of key block 0 -> registers, in flight under the softmax VALU |
//     P V of key block 0 (2 DB MFMAs) with key block 1's V fragments landing under it | P V of key block 1.
//     MFMA clusters are pure-register, so the other wave(s) of the SIMD run their softmax / fragment loads under them;
//   * K/V tiles come in by buffer_load ... lds: the buffer descriptor ends after key S-1, so the rows of a ragged last
//     tile read as ZERO in hardware (no clamp code, no NaN from stale memory under a zero probability) and the per-lane
//     offsets never change: a DMA instruction is {m0, voffset register, soffset = tile offset};
//   * ONE copy of the tile body: the ring slot is a run-time LDS base (an add per fragment base), so the loop is not
//     unrolled by parity and the body is instantiated twice (two live key blocks; one for the short last tile).
// VAR bit0: s_setprio 1 around the MFMA clusters; bit1: the exponentials of key block 1 are placed between the P V MFMAs
// of key block 0 (in-wave overlap); bit2: row sums by fp32 adds instead of v_dot2c_f32_bf16.
#include <type_traits>
#include <stdlib.h>
#include "common.hpp"
#include "kernels.hpp"
#include "attn_common.hpp"

// Timing-only ablation builds (tools/attn2_ablate.sh): -DATTN2_ABL=<mask> removes one cost at a time; results are wrong by
// construction.  1: no v_exp (the fma result is used as the probability), 2: no softmax arithmetic at all (scores are
// packed as they are), 4: no QK^T MFMAs, 8: no P V MFMAs, 16: no K/V DMA after the first tile, 32: no LDS fragment reads,
// 64: no per-tile barrier.
#ifndef ATTN2_ABL
#define ATTN2_ABL 0
#endif

namespace vllm {

namespace {

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ bf16x8_t lds_b128(uint32_t addr)
{
    bf16x8_t v;
    if constexpr ((ATTN2_ABL & 32) != 0) { asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr)); return v; }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ s16x4_t lds_tr_b64(uint32_t addr)
{
    s16x4_t v;
    if constexpr ((ATTN2_ABL & 32) != 0) { asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr)); return v; }
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// s_waitcnt lgkmcnt(CNT) with the fragments it releases as in/out operands: nothing that consumes them (nor a copy the
// register allocator makes to assemble an MFMA operand) can be placed above the wait.
template <int CNT, typename T, int N> __device__ __forceinline__ void wait_lgkm(T (&f)[N])
{
    static_assert(N == 4 || N == 8 || N == 16, "fragment batch size");
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is a 4-bit counter");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(CNT) : "memory");
    else if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                     : "n"(CNT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%16)"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                       "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15])
                     : "n"(CNT) : "memory");
}
// An MFMA (or any pure instruction) is not ordered against asm statements / scheduling fences by itself: instruction
// selection places it just above its first user.  Passing its result through an empty asm statement pins the producer
// above that point (and every consumer below it); no instruction is emitted.
__device__ __forceinline__ void pin(f32x16_t &x) { asm volatile("" : "+v"(x)); }

// K/V staging by buffer_load_dwordx4 ... lds.  A wave instruction moves 64 x 16 bytes = RPI rows of the tile to 1 KiB of
// LDS (lane-linear image: the bank swizzle is applied to the SOURCE chunk and undone by the fragment reads).
template <int D> struct Stage2 {
    static constexpr int CPR = D / 8;           // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;        // rows per wave instruction
    static constexpr int NI = KVBLK / RPI / 4;  // instructions per wave (4 waves per block)
};
template <int D, bool ISV> __device__ __forceinline__ void stage2_offsets(int ts, int wave, int lane, uint32_t (&vo)[Stage2<D>::NI])
{
    typedef Stage2<D> G;
#pragma unroll
    for (int s = 0; s < G::NI; ++s) {
        const int r = (wave * G::NI + s) * G::RPI + lane / G::CPR;
        const int c = (lane % G::CPR) ^ (ISV ? swz_v<D>(r) : swz_k<D>(r));
        vo[s] = (uint32_t)(r * ts + c * 8) * 2u;
    }
}
template <int D>
__device__ __forceinline__ void stage2(__amdgpu_buffer_rsrc_t rs, uint32_t tile_off, char *lds_tile, int wave,
                                       const uint32_t (&vo)[Stage2<D>::NI])
{
    typedef Stage2<D> G;
#pragma unroll
    for (int s = 0; s < G::NI; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds_tile + (wave * G::NI + s) * 1024),
                                                 16, (int)vo[s], (int)tile_off, 0, 0);
}

// Work.  XCD x owns the (batch, head) pairs bh = x, x + 8, ... (all query rows of a pair on one XCD: its K/V stay in that
// L2).  A block processes `a.nq` CONSECUTIVE query tiles (QBLK rows each) of ONE pair, one after the other, and the K/V
// tiles of these passes form ONE stream through the 2-slot LDS ring: the first tile of the next pass is requested during
// the last tile of the current one (same descriptors, tile offset wraps to 0).  With nq = all query tiles (the launch
// default for batches that fill the chip) a (batch, head) pair is ONE block: one dispatch, one cold start, and the
// nearly empty last query tile (S = 577 / 1025: 65 / 1 live rows) is the tail of a long block instead of a block of its
// own that occupies a CU slot for a full K/V sweep.  What that buys (profiles/r03_attn2_skeleton.txt: 3200 blocks of 2 / 4 /
// 10 / 19 KV tiles): a block costs 2.4 us + 1.37 us per tile at d = 64 and 5.8 us + 1.62 us per tile at d = 128 -- 15 - 17 %
// of a launch was per-block dispatch, Q / first-tile latency and store drain.
template <int D, int VAR>
__global__ __launch_bounds__(ATT_THREADS, D == 64 ? 3 : 2) void attn_fwd2_kernel(const AttnArgs a)
{
    constexpr bool PRIO = (VAR & 1) != 0, SPLIT = (VAR & 2) != 0, ADDSUM = (VAR & 4) != 0;
    constexpr int KS = D / 16;            // k-steps of the QK^T product = K fragments per 32-key block
    constexpr int DB = D / 32;            // 32-wide output blocks
    constexpr int NV = 4 * DB;            // V transpose-reads per 32-key block (2 halves of 16 keys x DB x {lo, hi})
    constexpr int TILE = KVBLK * D * 2;   // bytes per K or V tile
    constexpr float THR = 6.0f;           // deferred rescale (exp2 domain): P stays below 2^6
    constexpr bool QEARLY = D == 64;      // the next pass's Q rows are requested right after the last QK^T (registers allow it)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 slots][K | V]

    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();   // fragment reads address LDS by byte offset: no static LDS here
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- block -> (b, head, first query tile) ----
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int nchunk = (a.nqt + a.nq - 1) / a.nq;          // blocks per (b, head)
    const int bh = (sidx / nchunk) * 8 + xcd;
    if (bh >= a.B * a.H) return;
    const int qt0 = (sidx % nchunk) * a.nq;
    const int qt1 = min(a.nqt, qt0 + a.nq);
    const int b = bh / a.H, head = bh % a.H;

    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *kb_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *vb_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;
    // descriptors that end with the last element of key S-1: rows of a ragged last tile beyond it read as zero
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)kb_, 0, ((a.S - 1) * a.k_ts + D) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void *)vb_, 0, ((a.S - 1) * a.v_ts + D) * 2, 0x00020000);
    uint32_t kvo[Stage2<D>::NI], vvo[Stage2<D>::NI];
    stage2_offsets<D, false>(a.k_ts, wave, lane, kvo);
    stage2_offsets<D, true>(a.v_ts, wave, lane, vvo);
    const uint32_t k_step = (uint32_t)(KVBLK * a.k_ts * 2), v_step = (uint32_t)(KVBLK * a.v_ts * 2);   // bytes per tile

    const int nkt = (a.S + KVBLK - 1) / KVBLK;
    const bool short_tail = a.S - (nkt - 1) * KVBLK <= 32;
    const float c2 = a.scale_log2e;

    // Per-lane LDS offsets (the addresses of attn_fwd_kernel).  The swizzles are XORs of 16-byte chunk numbers, so the
    // offset of k-step ks / output block d is the offset of k-step 0 / block 0 XOR a constant; key-block parts are
    // instruction immediates; the ring slot is added per tile.
    const uint32_t kofs0 = (uint32_t)(l31 * (D * 2) + ((hh ^ swz_k<D>(l31)) << 4));
    uint32_t vofs0;
    {
        const int krow = 4 * hh + ((lane & 15) >> 2);
        const int c = 2 * ((lane >> 4) & 1) + (((lane & 15) & 3) >> 1);
        vofs0 = (uint32_t)(krow * (D * 2) + ((c ^ swz_v<D>(krow)) << 4) + (((lane & 15) & 1) << 3)) + (uint32_t)TILE;
    }

    // ---- Q fragments (B operand): lane (q = l31, hh) holds Q[q][16*ks + 8*hh .. +7] ----
    bf16x8_t qf[KS];
    auto load_q = [&](int qt) {
        const int q_row = qt * QBLK + wave * 32 + l31;
        const int q_ld = q_row < a.S ? q_row : a.S - 1;
        const uint16_t *qp = qb + (long)q_ld * a.q_ts + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t *>(qp + ks * 16);
    };
    // the tile after tile t of a pass: the next one, or the first of the next pass (block-uniform)
    auto stage_next = [&](int t, bool more_passes, uint32_t slot) {
        if (ATTN2_ABL & 16) return;
        const int tn = t + 1 < nkt ? t + 1 : 0;
        if (t + 1 < nkt || more_passes) {
            char *nx = smem + (2 * TILE - slot);
            stage2<D>(krs, (uint32_t)tn * k_step, nx, wave, kvo);
            stage2<D>(vrs, (uint32_t)tn * v_step, nx + TILE, wave, vvo);
        }
    };

    stage2<D>(krs, 0u, smem, wave, kvo);
    stage2<D>(vrs, 0u, smem + TILE, wave, vvo);
    if (qt0 * QBLK + wave * 32 < a.S) load_q(qt0);
    uint32_t slot = 0;               // ring slot of the tile being processed: 0 or 2 * TILE
    // Blocks of equal work that start together run in LOCKSTEP on a CU: their QK^T / softmax / P V phases coincide and
    // compete for the same pipe instead of filling each other's gaps (measured: all query tiles in one block, no stagger,
    // 101 vs 80 us at d = 64).  The k-th block a CU receives in the first wave of dispatches (blockIdx / #CUs) starts
    // a.stagger * k cycles late -- what the dispatcher's natural skew does for short blocks.
    if (a.stagger > 0) {
        const int k = blockIdx.x / a.n_cu;
        for (int i = 0; i < k * a.stagger; i += 64 * 16) __builtin_amdgcn_s_sleep(16);
    }

    for (int qt = qt0; qt < qt1; ++qt) {
        const bool more = qt + 1 < qt1;
        const bool live_wave = qt * QBLK + wave * 32 < a.S;
        if (!live_wave) {
            // a wave whose 32 query rows are all padding (the last query tile of S = 577 / 1025) only stages its share of the
            // K/V tiles and keeps the barrier count.  (Its own loop: carried through the loop below, the unused accumulators
            // of such a wave cost every live wave a 64-register copy per tile at the join.)
            for (int t = 0; t < nkt; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(ATTN2_ABL & 64)) __syncthreads();
                stage_next(t, more, slot);
                slot = 2 * TILE - slot;
            }
            continue;
        }
        f32x16_t o[DB];
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float m_run = -1.0e30f, l_run = 0.f;

        // exponentials + bf16 packing + row sum of ONE 32-key block
        typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
        auto exp_block = [&](const f32x16_t &s, uint32_t (&pk)[8], float &sum0, float &sum1) {
            const bf16x2v ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = (ATTN2_ABL & 1) ? fmaf(s[r], c2, -m_run) : __builtin_amdgcn_exp2f(fmaf(s[r], c2, -m_run));
                const float p1 = (ATTN2_ABL & 1) ? fmaf(s[r + 1], c2, -m_run) : __builtin_amdgcn_exp2f(fmaf(s[r + 1], c2, -m_run));
                const uint32_t w = pack_bf16x2(p0, p1);
                pk[r >> 1] = w;
                float &acc = ((r >> 1) & 1) ? sum1 : sum0;
                // sums of the bf16-ROUNDED probabilities (the ones the P V product uses): O / l normalises what was accumulated
                if constexpr (ADDSUM) acc += bf16lo_to_f32(w) + bf16hi_to_f32(w);
                else acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w), ones, acc, false);
            }
        };
        // O^T += V^T P^T for one 32-key block: 2 * DB MFMAs on fragments already in registers.
        // k-slots of step u: regs 8u..8u+7 <-> keys 16u + 4hh + {0..3, 8..11} of the block
        auto pv_block = [&](const uint32_t (&pk)[8], const s16x4_t (&hv)[NV]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 pw = {pk[4 * u], pk[4 * u + 1], pk[4 * u + 2], pk[4 * u + 3]};
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const s16x4_t v_lo = hv[(u * DB + d) * 2], v_hi = hv[(u * DB + d) * 2 + 1];
                    const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    if constexpr ((ATTN2_ABL & 8) != 0) { o[d][0] += __builtin_bit_cast(float, (int)vf[0] + (int)pf[0]); continue; }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
        };

        // One KV tile.  NKB: live 32-key blocks (1: the last tile holds <= 32 live keys -- S = 577 / 1025: the single
        // CLS-offset key).  Everything else about the tile is a run-time scalar.
        auto tile_step = [&](int t, auto nkb_) {
            constexpr int NKB = decltype(nkb_)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(ATTN2_ABL & 64)) __syncthreads();   // tile t landed for every wave; everyone is done reading the other slot
            stage_next(t, more, slot);
            const int k0 = t * KVBLK;
            uint32_t kbase = kofs0 + slot, vbase = vofs0 + slot;
            asm volatile("" : "+v"(kbase), "+v"(vbase));   // per tile: hoisted, these become KS + DB registers per slot
            // ---- K fragments of the whole tile -> registers, in the order the MFMAs consume them ----
            // The two key blocks' accumulators alternate (no MFMA waits for the one before it); the first half of the
            // k-steps is released by a counted wait while the second half is still in flight.
            bf16x8_t kfa[KS], kfb[KS];   // kfa: k-steps 0 .. KS/2-1 of {block 0, block 1}; kfb: the rest
            static_for<0, KS>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if constexpr (NKB == 2) kfa[i] = lds_b128<(i % 2) * 32 * (D * 2)>(kbase ^ (uint32_t)((i / 2) << 5));
                else kfa[i] = lds_b128<0>(kbase ^ (uint32_t)(i << 5));   // one live key block: all KS k-steps of block 0
            });
            if constexpr (NKB == 2) {
                static_for<0, KS>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    kfb[i] = lds_b128<(i % 2) * 32 * (D * 2)>(kbase ^ (uint32_t)((KS / 2 + i / 2) << 5));
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- S^T = K Q^T: dense MFMA cluster ----
            f32x16_t st0, st1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; }
            wait_lgkm<NKB == 2 ? KS : 0>(kfa);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            if constexpr (NKB == 2) {
#pragma unroll
                for (int i = 0; i < KS; i += 2) {
                    if constexpr ((ATTN2_ABL & 4) != 0) { st0[i] += __builtin_bit_cast(float, (int)kfa[i][0]); st1[i] += __builtin_bit_cast(float, (int)kfa[i + 1][0]); continue; }
                    st0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[i], qf[i / 2], st0, 0, 0, 0);
                    st1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[i + 1], qf[i / 2], st1, 0, 0, 0);
                }
                pin(st0); pin(st1);
                wait_lgkm<0>(kfb);
#pragma unroll
                for (int i = 0; i < KS; i += 2) {
                    if constexpr ((ATTN2_ABL & 4) != 0) { st0[i + 8] += __builtin_bit_cast(float, (int)kfb[i][0]); st1[i + 8] += __builtin_bit_cast(float, (int)kfb[i + 1][0]); continue; }
                    st0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[i], qf[KS / 2 + i / 2], st0, 0, 0, 0);
                    st1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[i + 1], qf[KS / 2 + i / 2], st1, 0, 0, 0);
                }
                pin(st0); pin(st1);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) st0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[ks], qf[ks], st0, 0, 0, 0);
                pin(st0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // (block-uniform) the pass's last QK^T is done and the Q registers are dead: d = 64 requests the next pass's Q rows
            // here, in flight under the rest of the tile
            if (QEARLY && t == nkt - 1 && more && (qt + 1) * QBLK + wave * 32 < a.S) load_q(qt + 1);
            // ---- V fragments of key block 0 -> registers (in flight under the softmax) ----
            // hv[(u * DB + d) * 2 + {0, 1}]: keys 16u + 4hh + i16/4 (+ 8) of the key block, chunk of output block d
            s16x4_t hv0[NV], hv1[NV];
            static_for<0, 2 * DB>([&](auto i_) {
                constexpr int i = decltype(i_)::value, u = i / DB, d = i % DB;
                hv0[i * 2] = lds_tr_b64<(16 * u) * (D * 2)>(vbase ^ (uint32_t)(d << 6));
                hv0[i * 2 + 1] = lds_tr_b64<(16 * u + 8) * (D * 2)>(vbase ^ (uint32_t)(d << 6));
            });
            __builtin_amdgcn_sched_barrier(0);
            // ---- online softmax (exp2 domain, deferred rescale) ----
            float ps0 = 0.f, ps1 = 0.f;
            uint32_t pk0[8], pk1[8];
            if constexpr ((ATTN2_ABL & 2) != 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) { pk0[r >> 1] = pack_bf16x2(st0[r], st0[r + 1]); pk1[r >> 1] = pack_bf16x2(st1[r], st1[r + 1]); }
            } else {
                if (k0 + KVBLK > a.S) {   // block-uniform: the ragged last tile
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        st0[r] = key < a.S ? st0[r] : -1.0e30f;
                        if (NKB == 2) st1[r] = key + 32 < a.S ? st1[r] : -1.0e30f;
                    }
                }
                float mx = -1.0e30f;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st0[r]);
                if constexpr (NKB == 2) {   // (its own chain: the two blocks' maxima are independent until here)
                    float mx1 = -1.0e30f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx1 = fmaxf(mx1, st1[r]);
                    mx = fmaxf(mx, mx1);
                }
                mx = halves_max(mx) * c2;                    // c2 > 0: max commutes with the scaling
                if (!__all(mx - m_run <= THR)) {             // wave-uniform; both halves of a query agree on mx
                    const float m_new = fmaxf(m_run, mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int d = 0; d < DB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                }
                exp_block(st0, pk0, ps0, ps1);
                if constexpr (NKB == 2 && !SPLIT) exp_block(st1, pk1, ps0, ps1);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- O^T += V^T P^T ----  (key block 1's V fragments are requested here and land under key block 0's MFMAs)
            wait_lgkm<0>(hv0);
            if constexpr (NKB == 2) {
                static_for<0, 2 * DB>([&](auto i_) {
                    constexpr int i = decltype(i_)::value, u = i / DB, d = i % DB;
                    hv1[i * 2] = lds_tr_b64<(32 + 16 * u) * (D * 2)>(vbase ^ (uint32_t)(d << 6));
                    hv1[i * 2 + 1] = lds_tr_b64<(32 + 16 * u + 8) * (D * 2)>(vbase ^ (uint32_t)(d << 6));
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            pv_block(pk0, hv0);
            if constexpr (NKB == 2) {
                if constexpr (SPLIT && !(ATTN2_ABL & 2)) {
                    // key block 0's MFMAs and key block 1's exponentials in ONE scheduling region: a few VALU per MFMA gap
                    exp_block(st1, pk1, ps0, ps1);
#pragma unroll
                    for (int i = 0; i < 2 * DB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 56 / (2 * DB), 0);        // its share of the ~56 VALU / TRANS
                    }
                }
#pragma unroll
                for (int d = 0; d < DB; ++d) pin(o[d]);
                if constexpr (SPLIT) __builtin_amdgcn_sched_barrier(0);
                wait_lgkm<0>(hv1);
                pv_block(pk1, hv1);
            }
#pragma unroll
            for (int d = 0; d < DB; ++d) pin(o[d]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            l_run += ps0 + ps1;
            __builtin_amdgcn_sched_barrier(0);
            slot = 2 * TILE - slot;
        };
        constexpr std::integral_constant<int, 1> ONE_BLOCK{};
        constexpr std::integral_constant<int, 2> TWO_BLOCKS{};
        for (int t = 0; t < nkt - 1; ++t) tile_step(t, TWO_BLOCKS);
        if (short_tail) tile_step(nkt - 1, ONE_BLOCK); else tile_step(nkt - 1, TWO_BLOCKS);

        // ---- finalize: O / l ; lane holds d = 32*db + 8*(r>>2) + 4*hh + (r&3) of query l31 ----
        // (d = 128: the next pass's Q rows are requested first -- every fragment register is free now -- and land under
        //  the normalisation and the stores)
        if (!QEARLY && more && (qt + 1) * QBLK + wave * 32 < a.S) load_q(qt + 1);
        const int q_row = qt * QBLK + wave * 32 + l31;
        const float l_tot = halves_sum(l_run);
        const float inv = 1.0f / l_tot;
        if (q_row < a.S) {
            uint16_t *orow = a.out + (((long)b * a.S + q_row) * a.H + head) * D;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    uint2_t w;
                    w.x = pack_bf16x2(o[d][4 * rq] * inv, o[d][4 * rq + 1] * inv);
                    w.y = pack_bf16x2(o[d][4 * rq + 2] * inv, o[d][4 * rq + 3] * inv);
                    *reinterpret_cast<uint2_t *>(orow + d * 32 + 8 * rq + 4 * hh) = w;
                }
        }
    }
}

}  // namespace

int attn_fwd2_launch(AttnArgs a, int D, int var2, hipStream_t st)
{
    // the buffer descriptors address one (batch, head) slab with 32-bit byte offsets
    VLLM_REQUIRE(((long)a.S * a.k_ts + D) * 2 < (1l << 31) && ((long)a.S * a.v_ts + D) * 2 < (1l << 31),
                 "attn: a (batch, head) K/V slab must stay below 2 GiB (S=%d, token stride %d)", a.S, a.k_ts);
    const long groups = ((long)a.B * a.H + 7) / 8;
    // Query tiles per block: ONE (default).  Letting a block walk ALL query tiles of its (batch, head) pair (var2 bit 3;
    // one dispatch and one cold start per pair, no nearly empty tail block) measured SLOWER on MI355X: 101 vs 80 us at
    // 40 x 16 x S577 x d64 and 823 vs 669 us at 40 x 25 x S1025 x d128 (profiles/r03_attn2_nq_and_stagger.txt) -- the pair's
    // K/V is then re-read 5 / 9 times over a long period by one CU instead of being shared, within a few microseconds, by
    // the 5 / 9 blocks of the pair running side by side on the XCD: the concurrently live pairs (64-96 per XCD x 148-524 KB)
    // no longer fit the 4 MiB L2.  A start skew between the blocks of a CU (a.stagger) changed nothing (102-104 us).
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    a.nq = (var2 & 8) ? a.nqt : 1;
    // stagger (cycles between the blocks of a CU): a third / half of a tile's time; VLLM_ATTN_STAGGER overrides (0 = off)
    static const int stagger_env = [] { const char *e = getenv("VLLM_ATTN_STAGGER"); return e ? atoi(e) : -1; }();
    a.n_cu = cus;
    a.stagger = a.nq > 1 ? (stagger_env >= 0 ? stagger_env : (D == 64 ? 1100 : 2000)) : 0;
    const long nblk = groups * 8 * ((a.nqt + a.nq - 1) / a.nq);
    const dim3 grid((unsigned)nblk), block(ATT_THREADS);
    const size_t lds = 4 * (size_t)KVBLK * D * 2;
#define LA2(DD, V) VLLM_LAUNCH((attn_fwd2_kernel<DD, V>), grid, block, lds, st, a)
#define LV2(DD) do { switch (var2 & 7) { case 0: LA2(DD, 0); break; case 1: LA2(DD, 1); break; case 2: LA2(DD, 2); break; \
    case 3: LA2(DD, 3); break; case 4: LA2(DD, 4); break; case 5: LA2(DD, 5); break; case 6: LA2(DD, 6); break; \
    default: LA2(DD, 7); } } while (0)
    if (D == 64) LV2(64); else LV2(128);
#undef LV2
#undef LA2
    VLLM_CHECK_LAUNCH("attn_fwd2_kernel");
    return VLLM_OK;
}

#ifdef ATTN2_ABL_ENTRY
// test entry of the ablation builds (tools/attn2_ablate.py)
extern "C" int attn2_abl_run(const uint16_t *qkv, uint16_t *out, int B, int S, int H, int D, float scale, int var2, void *stream)
{
    AttnArgs a;
    const long C = (long)H * D;
    a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C; a.out = out;
    a.q_bs = a.k_bs = a.v_bs = (long)S * 3 * C;
    a.q_ts = a.k_ts = a.v_ts = (int)(3 * C);
    a.q_hs = a.k_hs = a.v_hs = D;
    a.B = B; a.S = S; a.H = H; a.nqt = (S + QBLK - 1) / QBLK; a.row0 = 0; a.no_trim = 0;
    a.scale_log2e = scale * 1.4426950408889634f;
    return attn_fwd2_launch(a, D, var2, (hipStream_t)stream);
}
void set_error(const char *, ...) {}
#endif

}  // namespace vllm
