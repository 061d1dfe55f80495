// Fused (flash-style) multi-head self-attention forward for the ViT tiles, bf16 in/out, fp32 softmax/accumulate.
//
// Replaces  FlashAttention.forward / flash_attn_varlen_qkvpacked_func
//             (VisionLLMv2/visionllmv2/model/internvit/flash_attention.py:30-75; causal=False, dropout 0,
//              softmax_scale = d^-0.5) and InternAttention._naive_attn's (q*scale)@k^T -> softmax -> @v
//             (modeling_intern_vit.py:136-140); same math for CLIP's eager attention.
// Tiles never attend to each other, so the attention "window" is one tile: S = 577 (336^2) or 1025 (448^2).
//
// gfx950 design (wave64, v_mfma_f32_32x32x16_bf16):
//   * block = 4 waves = 128 query rows of one (tile, head); each wave owns 32 query rows, Q lives in registers;
//   * K/V tiles of 64 keys are staged by LDS-DMA (global_load_lds_dwordx4) into a 2-stage ring; the LDS image is
//     lane-linear, so the bank swizzles are applied to the per-lane SOURCE address and undone on the read side;
//   * scores are computed TRANSPOSED (S^T = K Q^T): each lane then holds 32 scores of ONE query row, so the
//     softmax row reductions are in-register plus a single lane<->lane+32 exchange, and P (converted in place
//     to bf16) already has the MFMA B-operand layout for O^T += V^T P^T -- no P round trip through LDS;
//   * V stays row-major in LDS and is read with the hardware transpose ds_read_b64_tr_b16 (layout verified on
//     the device by tools/probes/probe.hip);
//   * online softmax in the exp2 domain (scale*log2(e) folded into one FMA), key tail masked in the last tile;
//   * (tile, head) -> XCD mapping keeps all query blocks of a head on one XCD (K/V re-reads hit that L2).
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "attn_common.hpp"

namespace vllm {

#ifndef ATT_EPI_DEFAULT     // 64: the automatic schedule stores O through LDS (round 5: ViT-L 82.0 -> 79.1 us, InternViT-6B 657 -> 628 us at 40 tiles,
                            // bit-identical outputs, profiles/r05_attn_epilogue.txt); 0: straight from the accumulators (attn_variant 2)
#define ATT_EPI_DEFAULT 64
#endif

// K/V staging.  The per-lane part of every source address (row-in-tile * token stride + swizzled 16-byte chunk) does not
// change from tile to tile: it is computed ONCE (kv_lane_offsets) and each tile's LDS-DMA is then
// (uniform tile base in SGPRs) + (that 32-bit lane offset) -> the saddr form of global_load_lds with a uniform LDS
// destination.  Left per tile, the 64-bit row multiply / clamp / readfirstlane chain was ~12 VALU per tile on a kernel
// whose VALU issue is the bound.  Only the last tile of a ragged S clamps rows (keys >= S re-read key S-1; masked later).
template <int D, int WPB = 4> struct KvStage {   // WPB = waves per block sharing the staging work
    static constexpr int CPR = D / 8;           // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;        // rows per wave instruction (1 KiB)
    static constexpr int NI = KVBLK / RPI / WPB;  // instructions per wave
};

template <int D, bool ISV, int WPB = 4>
__device__ __forceinline__ void kv_lane_offsets(int ts, int wave, int lane, uint32_t (&vo)[KvStage<D, WPB>::NI])
{
    typedef KvStage<D, WPB> G;
#pragma unroll
    for (int s = 0; s < G::NI; ++s) {
        const int r = (wave * G::NI + s) * G::RPI + lane / G::CPR;
        const int c = (lane % G::CPR) ^ (ISV ? swz_v<D>(r) : swz_k<D>(r));
        vo[s] = (uint32_t)(r * ts + c * 8) * 2u;
    }
}

template <int D, bool ISV, bool RAGGED = true, int WPB = 4>
__device__ __forceinline__ void stage_kv(const uint16_t *__restrict__ base, int ts, int k0, int S, char *lds_tile,
                                         int wave, int lane, const uint32_t (&vo)[KvStage<D, WPB>::NI])
{
    typedef KvStage<D, WPB> G;
    uint32_t off[G::NI];
#pragma unroll
    for (int s = 0; s < G::NI; ++s) off[s] = vo[s];
    if (RAGGED && k0 + KVBLK > S) {   // block-uniform: ragged last tile, rows past the last key re-read key S-1
#pragma unroll
        for (int s = 0; s < G::NI; ++s) {
            const int r = (wave * G::NI + s) * G::RPI + lane / G::CPR;
            const int c = (lane % G::CPR) ^ (ISV ? swz_v<D>(r) : swz_k<D>(r));   // the LDS image keeps row r's swizzle
            const int rr = k0 + r < S ? r : S - 1 - k0;
            off[s] = (uint32_t)(rr * ts + c * 8) * 2u;
        }
    }
    // ONE load site per instruction: (uniform tile base) + (32-bit lane offset) selects the saddr form, uniform LDS address.
    // The empty asm keeps the zero-extension of the offset from being hoisted out of the tile loop as a 64-bit register
    // pair (which turns every DMA back into a 64-bit VALU add + vaddr form).
    const char *tile = reinterpret_cast<const char *>(base + (long)k0 * ts);
#pragma unroll
    for (int s = 0; s < G::NI; ++s) asm volatile("" : "+v"(off[s]));
#pragma unroll
    for (int s = 0; s < G::NI; ++s)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tile + off[s]),
                                         (__attribute__((address_space(3))) void *)(lds_tile + (wave * G::NI + s) * 1024), 16,
                                         0, 0);
}

// VAR bit0: software-pipelined K (scores of tile t+1 next to the softmax of tile t); bit1: deferred rescale;
// bit2: s_setprio(1) around the MFMA clusters; bit3: V transpose-reads issued as inline asm BEFORE the softmax (the
// compiler treats the tr-read builtin as 'may alias the LDS-DMA in flight' and puts s_waitcnt vmcnt(0) in front of it,
// which drains the next tile's prefetch in the middle of every iteration).
// EPI: 0 = every lane stores its 8-byte pieces straight from the accumulator layout (16 bytes per row and instruction: 32 partial
// lines per store); 1 (round 5, the review's item 3-i / guide T21's LDS form) = the block's O tile goes through the (by then idle)
// K/V ring and leaves as WHOLE ROWS, 16 bytes per lane, 8 (d = 64) or 4 (d = 128) full rows per instruction.
template <int D, int VAR, bool F16 = false, int EPI = 0>
__global__ __launch_bounds__(ATT_THREADS, (D == 64 && !(VAR & 9)) ? 4 : 2) void attn_fwd_kernel(const AttnArgs a)
{
    constexpr bool PIPE = (VAR & 1) != 0, DEFER = (VAR & 2) != 0, PRIO = (VAR & 4) != 0, ASMTR = (VAR & 8) != 0;
    constexpr int KS = D / 16;            // k-steps of the QK^T product
    constexpr int DB = D / 32;            // 32-wide output blocks
    constexpr int TILE = KVBLK * D * 2;   // bytes per K or V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K | V]

    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();   // fragment reads address LDS by byte offset: no static LDS here
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave index in an SGPR
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- block -> (b, head, q tile): all q tiles of a (b, head) on one XCD ----
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int bh = (sidx / a.nqt) * 8 + xcd;
    const int qt = sidx % a.nqt;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, head = bh % a.H;

    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *kb_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *vb_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;

    uint32_t kvo[KvStage<D>::NI], vvo[KvStage<D>::NI];
    kv_lane_offsets<D, false>(a.k_ts, wave, lane, kvo);
    kv_lane_offsets<D, true>(a.v_ts, wave, lane, vvo);

    // ---- Q fragments (B operand): lane (q = l31, hh) holds Q[q][16*ks + 8*hh .. +7] ----
    const int q_row = qt * QBLK + wave * 32 + l31;
    const int q_ld = q_row < a.S ? q_row : a.S - 1;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8_t *>(qb + (long)q_ld * a.q_ts + ks * 16 + hh * 8);

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;
    const float c2 = a.scale_log2e;
    const int nkt = (a.S + KVBLK - 1) / KVBLK;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;

    // Loop-invariant LDS offsets.  swz_k / swz_v only look at key bits that come from the lane (the block constants
    // 32*kb, 16*u, +8 do not reach them), so every fragment address is (tile base) + (one of these) + (an immediate):
    // left to the compiler the XOR swizzle was re-evaluated per read, ~45 v_add per KV tile on a VALU-bound kernel.
    int kofs[KS], vofs[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kofs[ks] = l31 * (D * 2) + (((2 * ks + hh) ^ swz_k<D>(l31)) << 4);
    {
        const int krow = 4 * hh + ((lane & 15) >> 2);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int c = d * 4 + 2 * ((lane >> 4) & 1) + (((lane & 15) & 3) >> 1);
            vofs[d] = krow * (D * 2) + ((c ^ swz_v<D>(krow)) << 4) + (((lane & 15) & 1) << 3);
        }
    }
    // S^T = K Q^T for one staged K tile: two 32-key blocks, KS chained MFMAs each
    // NKB = 1: the last tile holds <= 32 live keys (S = 577 / 1025: the single CLS-offset key) -- second key block skipped
    auto qk = [&](uint32_t ks_, f32x16_t (&st)[2], auto nkb_) {
        constexpr int NKB = decltype(nkb_)::value;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *(const __attribute__((address_space(3))) bf16x8_t *)(uintptr_t)(ks_ + kofs[ks] + kb * 32 * (D * 2));
                st[kb] = mfma16<F16>(kf, qf[ks], st[kb]);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // online softmax of one score tile (raw scores; scale*log2e folded into the exp2 argument) + O^T += V^T P^T.
    // Lane holds keys kb*32 + (r&3) + 8*(r>>2) + 4*hh of query l31.  Rescale of O / l is DEFERRED while the running
    // max grows by less than THR (exp2 domain): P is then bounded by 2^THR instead of 1 (fp32 accumulation).
    auto softmax_pv = [&](f32x16_t (&st)[2], uint32_t vs_, int k0, auto nkb_, auto mask_) {
        constexpr bool MASK = decltype(mask_)::value;   // false: the tile is known to be full (steady-state steps)
        constexpr int NKB = decltype(nkb_)::value;
        constexpr float THR = DEFER ? 6.0f : 0.0f;
        constexpr int NHOIST = ASMTR ? 16 : 1;          // tr-reads hoisted above the softmax: steps (kb,u) x d blocks
        s16x4_t hv[NHOIST];                              // D=64: all 16 reads of the tile; D=128: the kb=0 half
        if constexpr (ASMTR) {
            const uint32_t vbase = vs_;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // i -> (step, d, lo/hi) in the order the PV loop consumes them
                const int per_step = 2 * DB;
                const int step = i / per_step, rem = i % per_step, d = rem >> 1, hi_ = rem & 1;
                const int kb = step >> 1, u = step & 1;
                const int key = kb * 32 + 16 * u + 4 * hh + (i16 >> 2) + 8 * hi_;
                const int c = d * 4 + 2 * g1 + ((i16 & 3) >> 1);
                const uint32_t addr = vbase + key * (D * 2) + ((c ^ swz_v<D>(key)) << 4) + ((i16 & 1) << 3);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hv[i]) : "v"(addr));
            }
        }
        float mx = -1.0e30f;
        if (MASK && k0 + KVBLK > a.S) {   // tail tile: mask keys >= S (block-uniform branch)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    st[kb][r] = key < a.S ? st[kb][r] : -1.0e30f;
                }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = halves_max(mx) * c2;                        // c2 > 0: max commutes with the scaling
        if (!DEFER || !__all(mx - m_run <= THR)) {          // wave-uniform; both halves of a query agree on mx
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        // Row sums are taken from the bf16-ROUNDED probabilities (the ones the PV product uses) with one v_dot2c_f32_bf16
        // against (1, 1) per pair: 16 VALU per tile instead of 31 adds, and O / l normalises exactly what was accumulated.
        float psum[2] = {0.f, 0.f};
        uint32_t pk[2][8];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(st[kb][r], c2, -m_run));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(st[kb][r + 1], c2, -m_run));
                const uint32_t w = pack16x2<F16>(p0, p1);
                pk[kb][r >> 1] = w;
                psum[(r >> 1) & 1] = dot2_ones<F16>(w, psum[(r >> 1) & 1]);
            }
        l_run += psum[0] + psum[1];
        // O^T += V^T P^T ; k-slots of step (kb,u): regs 8u..8u+7 <-> keys 32kb+16u+4hh+{0..3, 8..11}
        if constexpr (ASMTR) {
            // the hoisted reads have landed (cdna guide 5.7, form iii).  The registers are in/out operands of the wait so
            // that no compiler-made copy of them (tuple assembly for the MFMA operand) can be placed above it.
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]), "+v"(hv[6]), "+v"(hv[7]),
                           "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]), "+v"(hv[13]), "+v"(hv[14]),
                           "+v"(hv[15])
                         :
                         : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 pw = {pk[kb][4 * u], pk[kb][4 * u + 1], pk[kb][4 * u + 2], pk[kb][4 * u + 3]};
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
                // lane (key 4hh + i16/4 (+8 for v_hi), chunk 4d + 2g1 + (i16&3)/2) of block (kb, u): vofs[d] + immediates
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int hidx = ((kb * 2 + u) * DB + d) * 2;   // position in the hoisted set (if it is in it)
                    s16x4_t v_lo, v_hi;
                    if (ASMTR && hidx + 1 < NHOIST) {
                        v_lo = hv[hidx < NHOIST ? hidx : 0];
                        v_hi = hv[hidx + 1 < NHOIST ? hidx + 1 : 0];
                    } else {
                        const int blk = (kb * 32 + 16 * u) * (D * 2);   // immediate; key2 = key1 + 8 shares the swizzle
                        v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk));
                        v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk + 8 * (D * 2)));
                    }
                    const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    o[d] = mfma16<F16>(vf, pf, o[d]);
                }
            }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    constexpr std::integral_constant<bool, true> CTRUE{};
    constexpr std::integral_constant<int, 2> FULL{};
    constexpr std::integral_constant<int, 1> HALF{};
    if constexpr (PIPE) {
        // LDS: [K slot 0 | V slot 0 | K slot 1 | V slot 1].  K runs ONE tile ahead of V: iteration t computes the scores of
        // tile t+1 (MFMA) next to the softmax of tile t (VALU) -- independent streams inside one wave.
        auto kslot = [&](int t) { return smem + (t & 1) * 2 * TILE; };
        auto vslot = [&](int t) { return smem + (t & 1) * 2 * TILE + TILE; };
        auto kaddr = [&](int t) { return (uint32_t)((t & 1) * 2 * TILE); };   // LDS byte addresses (dynamic LDS starts at 0)
        auto vaddr = [&](int t) { return (uint32_t)((t & 1) * 2 * TILE + TILE); };
        stage_kv<D, false>(kb_, a.k_ts, 0, a.S, kslot(0), wave, lane, kvo);
        stage_kv<D, true>(vb_, a.v_ts, 0, a.S, vslot(0), wave, lane, vvo);
        if (nkt > 1) stage_kv<D, false>(kb_, a.k_ts, KVBLK, a.S, kslot(1), wave, lane, kvo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16_t sA[2], sB[2];
        qk(kaddr(0), sA, FULL);

        auto iteration = [&](int t, f32x16_t (&cur)[2], f32x16_t (&nxt)[2]) {
            // K_{t+1}, V_t (issued one iteration ago) have landed for every wave, and every wave is done reading the
            // slots of K_t / V_{t-1} that are refilled below (t = 0: the prologue's K_0 reads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < nkt) stage_kv<D, false>(kb_, a.k_ts, (t + 2) * KVBLK, a.S, kslot(t), wave, lane, kvo);
            if (t + 1 < nkt) stage_kv<D, true>(vb_, a.v_ts, (t + 1) * KVBLK, a.S, vslot(t + 1), wave, lane, vvo);
            if (t + 1 < nkt) qk(kaddr(t + 1), nxt, FULL);
            softmax_pv(cur, vaddr(t), t * KVBLK, FULL, CTRUE);
        };
        for (int t = 0; t < nkt; t += 2) {
            iteration(t, sA, sB);
            if (t + 1 < nkt) iteration(t + 1, sB, sA);
        }
    } else {
        // plain schedule: K_t and V_t staged together one tile ahead; QK -> softmax -> PV in sequence.  Waves whose 32
        // query rows are all padding (S = 577: wave 3 of the last query block) only stage and synchronise.
        const bool live_wave = a.no_trim || qt * QBLK + wave * 32 < a.S;
        const bool short_tail = !a.no_trim && a.S - (nkt - 1) * KVBLK <= 32;
        stage_kv<D, false>(kb_, a.k_ts, 0, a.S, smem, wave, lane, kvo);
        stage_kv<D, true>(vb_, a.v_ts, 0, a.S, smem + TILE, wave, lane, vvo);
        // One tile: wait for it, start the next one's DMA, QK -> softmax -> PV.  STAGE 0/1: the ring slot is a compile-time
        // constant (LDS fragment addresses become register + immediate); -1: taken from t.  NEXT 0: nothing to stage,
        // 1: the next tile is a full one (no row clamp code), 2: it may be the ragged last tile.
        auto tile_step = [&](int t, auto nkb_, auto stage_, auto next_) {
            constexpr int STAGE = decltype(stage_)::value, NEXT = decltype(next_)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // tile t landed for every wave; everyone is done reading the other stage
            const int slot = STAGE >= 0 ? STAGE : (t & 1);
            const uint32_t ks_ = (uint32_t)(slot * 2 * TILE);   // LDS byte address of the K slot (dynamic LDS starts at 0)
            if constexpr (NEXT != 0) {
                char *nx = smem + (slot ^ 1) * 2 * TILE;
                stage_kv<D, false, NEXT == 2>(kb_, a.k_ts, (t + 1) * KVBLK, a.S, nx, wave, lane, kvo);
                stage_kv<D, true, NEXT == 2>(vb_, a.v_ts, (t + 1) * KVBLK, a.S, nx + TILE, wave, lane, vvo);
            }
            if (live_wave) {
                f32x16_t st[2];
                qk(ks_, st, nkb_);
                softmax_pv(st, ks_ + TILE, t * KVBLK, nkb_, std::integral_constant<bool, NEXT != 1>{});
            }
        };
        constexpr std::integral_constant<int, 0> C0{};
        constexpr std::integral_constant<int, 1> C1{};
        constexpr std::integral_constant<int, 2> C2{};
        constexpr std::integral_constant<int, -1> CDYN{};
        // steady state: steps whose next tile is full, unrolled by the ring parity; then the step that stages the (possibly
        // ragged) last tile; then the last tile itself (peeled: one copy of the body per loop, register pressure)
        const int n_main = nkt - 2;
        for (int t = 0; t < n_main; t += 2) {
            tile_step(t, FULL, C0, C1);
            if (t + 1 < n_main) tile_step(t + 1, FULL, C1, C1);
        }
        if (nkt >= 2) tile_step(nkt - 2, FULL, CDYN, C2);
        if (short_tail) tile_step(nkt - 1, HALF, CDYN, C0); else tile_step(nkt - 1, FULL, CDYN, C0);
    }

    // ---- finalize: O / l ; lane holds d = 32*db + 8*(r>>2) + 4*hh + (r&3) of query l31 ----
    const float l_tot = halves_sum(l_run);
    const float inv = 1.0f / l_tot;
    if constexpr (EPI == 1) {
        // Rows through LDS.  The ring is dead once EVERY wave has left its last tile (one extra barrier per block); a wave then
        // uses only ITS 32 rows x D x 2 bytes of it: no further synchronisation.  Row pitch D * 2 bytes (128 / 256: a multiple of
        // the 256-byte bank span), so the 16-byte chunk index is XORed with the row (8 / 16 chunks per row): the 8-byte writes of a
        // 16-lane group (16 rows, one chunk column) and the 16-byte reads (8 / 4 rows, every chunk) then spread over the banks.
        constexpr int CPRO = D / 8;                   // 16-byte chunks per output row
        __syncthreads();
        char *wbase = smem + wave * (32 * D * 2);
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2_t w;
                w.x = pack16x2<F16>(o[d][4 * rq] * inv, o[d][4 * rq + 1] * inv);
                w.y = pack16x2<F16>(o[d][4 * rq + 2] * inv, o[d][4 * rq + 3] * inv);
                const int chunk = d * 4 + rq;         // columns 8 * chunk .. + 7; this lane's half: 4 * hh
                *reinterpret_cast<uint2_t *>(wbase + l31 * (D * 2) + ((chunk ^ (l31 & (CPRO - 1))) << 4) + hh * 8) = w;
            }
        // (same wave wrote what it reads: the LDS executes a wave's operations in order; the compiler's own lgkmcnt covers the data)
        constexpr int RPI = 64 / CPRO;                // rows per store instruction
        const int rr = lane / CPRO, cc = lane % CPRO;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + rr;
            const uint4_t v = *reinterpret_cast<const uint4_t *>(wbase + row * (D * 2) + ((cc ^ (row & (CPRO - 1))) << 4));
            const int qr = qt * QBLK + wave * 32 + row;
            if (qr < a.S) *reinterpret_cast<uint4_t *>(a.out + (((long)b * a.S + qr) * a.H + head) * D + cc * 8) = v;
        }
    } else if (q_row < a.S) {
        uint16_t *orow = a.out + (((long)b * a.S + q_row) * a.H + head) * D;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2_t w;
                w.x = pack16x2<F16>(o[d][4 * rq] * inv, o[d][4 * rq + 1] * inv);
                w.y = pack16x2<F16>(o[d][4 * rq + 2] * inv, o[d][4 * rq + 3] * inv);
                *reinterpret_cast<uint2_t *>(orow + d * 32 + 8 * rq + 4 * hh) = w;
            }
    }
}


int attn_fwd_launch(AttnArgs a, int D, hipStream_t st)
{
    VLLM_REQUIRE(a.B >= 0 && a.S > 0 && a.H > 0, "attn: bad dims B=%d S=%d H=%d", a.B, a.S, a.H);
    VLLM_REQUIRE(D == 64 || D == 128, "attn: head_dim %d not supported (64 or 128)", D);
    if (a.B == 0) return VLLM_OK;
    VLLM_REQUIRE(a.q && a.k && a.v && a.out, "attn: null pointer");
    VLLM_REQUIRE(aligned16(a.q) && aligned16(a.k) && aligned16(a.v) && (reinterpret_cast<uintptr_t>(a.out) & 7u) == 0 &&
                     a.q_ts % 8 == 0 && a.k_ts % 8 == 0 && a.v_ts % 8 == 0 && a.q_hs % 8 == 0 && a.k_hs % 8 == 0 &&
                     a.v_hs % 8 == 0 && a.q_bs % 8 == 0 && a.k_bs % 8 == 0 && a.v_bs % 8 == 0,
                 "attn: q/k/v must be 16-byte aligned with strides multiple of 8 elements");
    a.nqt = (a.S + QBLK - 1) / QBLK;
    a.row0 = 0;
    const long groups = ((long)a.B * a.H + 7) / 8;
    const dim3 grid((unsigned)(groups * 8 * a.nqt)), block(ATT_THREADS);
    const size_t lds = 4 * (size_t)KVBLK * D * 2;
    // attn_variant: 32 = automatic; bits 0-3 are attn_fwd_kernel's VAR, bit 4 switches the padding trim off.  (Schedule 2 -- the
    // hand-placed instruction stream of round 3, exactly as fast -- is tools/experiments/attn2.hip since round 4.)
    int var = attn_variant();
    if (var & 32) var = 2 | ATT_EPI_DEFAULT;
    a.no_trim = (var >> 4) & 1;
    const bool epi_lds = (var >> 6) & 1;      // bit 6: O through LDS, whole-row stores (needs 16-byte aligned output rows)
    var &= 15;
    if (epi_lds && var == 2 && aligned16(a.out)) {
        if (a.f16) {
            if (D == 64) VLLM_LAUNCH((attn_fwd_kernel<64, 2, true, 1>), grid, block, lds, st, a);
            else VLLM_LAUNCH((attn_fwd_kernel<128, 2, true, 1>), grid, block, lds, st, a);
        } else {
            if (D == 64) VLLM_LAUNCH((attn_fwd_kernel<64, 2, false, 1>), grid, block, lds, st, a);
            else VLLM_LAUNCH((attn_fwd_kernel<128, 2, false, 1>), grid, block, lds, st, a);
        }
        VLLM_CHECK_LAUNCH("attn_fwd_kernel");
        return VLLM_OK;
    }
#define LA(DD, V) VLLM_LAUNCH((attn_fwd_kernel<DD, V>), grid, block, lds, st, a)
#define LV(DD) do { switch (var) { case 0: LA(DD, 0); break; case 2: LA(DD, 2); break; case 6: LA(DD, 6); break; \
    case 8: LA(DD, 8); break; case 10: LA(DD, 10); break; case 14: LA(DD, 14); break; case 3: LA(DD, 3); break; \
    default: LA(DD, 2); } } while (0)
    if (a.f16) {   // IEEE half: the default schedule only
        if (D == 64) VLLM_LAUNCH((attn_fwd_kernel<64, 2, true>), grid, block, lds, st, a);
        else VLLM_LAUNCH((attn_fwd_kernel<128, 2, true>), grid, block, lds, st, a);
    } else if (D == 64) LV(64); else LV(128);
#undef LV
#undef LA
    VLLM_CHECK_LAUNCH("attn_fwd_kernel");
    return VLLM_OK;
}

}  // namespace vllm

using namespace vllm;

// B4: FlashAttention.forward(qkv[B,S,3,H,D]) -> out[B,S,H,D]   (flash_attention.py:30-75)
extern "C" int vllm_attn_fwd_qkvpacked_bf16(const uint16_t *qkv, uint16_t *out, int B, int S, int H, int D,
                                            float softmax_scale, vllm_stream_t stream)
{
    AttnArgs a;
    const long C = (long)H * D;
    a.q = qkv; a.k = qkv ? qkv + C : nullptr; a.v = qkv ? qkv + 2 * C : nullptr; a.out = out;
    a.q_bs = a.k_bs = a.v_bs = (long)S * 3 * C;
    a.q_ts = a.k_ts = a.v_ts = (int)(3 * C);
    a.q_hs = a.k_hs = a.v_hs = D;
    a.B = B; a.S = S; a.H = H; a.nqt = 0;
    a.scale_log2e = softmax_scale * 1.4426950408889634f;
    return attn_fwd_launch(a, D, (hipStream_t)stream);
}
// The same for IEEE-half qkv (the reference's FlashAttention accepts fp16 and bf16, flash_attention.py:39-41).
extern "C" int vllm_attn_fwd_qkvpacked_f16(const uint16_t *qkv, uint16_t *out, int B, int S, int H, int D,
                                           float softmax_scale, vllm_stream_t stream)
{
    AttnArgs a;
    const long C = (long)H * D;
    a.q = qkv; a.k = qkv ? qkv + C : nullptr; a.v = qkv ? qkv + 2 * C : nullptr; a.out = out;
    a.q_bs = a.k_bs = a.v_bs = (long)S * 3 * C;
    a.q_ts = a.k_ts = a.v_ts = (int)(3 * C);
    a.q_hs = a.k_hs = a.v_hs = D;
    a.B = B; a.S = S; a.H = H; a.nqt = 0;
    a.scale_log2e = softmax_scale * 1.4426950408889634f;
    a.f16 = 1;
    return attn_fwd_launch(a, D, (hipStream_t)stream);
}
