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

// DEFER: the running-max rescale is skipped while the maximum grows by less than 2^6 (cdna guide: defer-max).
// EPI: 0 = every lane stores its 8-byte pieces straight from the accumulator layout (16 bytes per row and instruction: 32 partial
// lines per store); 1 (round 5, guide T21's LDS form) = the block's O tile goes through the (by then idle) K/V ring and leaves
// as WHOLE ROWS, 16 bytes per lane, 8 (d = 64) or 4 (d = 128) full rows per instruction.
// (Rounds 2-5 carried three more schedule bits -- software-pipelined K, s_setprio around the MFMA clusters, hoisted asm
//  transpose reads; each measured as fast or slower on both shapes, none was the default: removed in round 6, history keeps them.)
template <int D, bool DEFER, bool F16 = false, int EPI = 0>
__global__ __launch_bounds__(ATT_THREADS, D == 64 ? 4 : 2) void attn_fwd_kernel(const AttnArgs a)
{
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
    const int Sk = a.S - a.kx;            // keys of the tile loop: kx .. S-1

    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *k0_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *v0_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;
    const uint16_t *kb_ = k0_ + (long)a.kx * a.k_ts;
    const uint16_t *vb_ = v0_ + (long)a.kx * a.v_ts;

    uint32_t kvo[KvStage<D>::NI], vvo[KvStage<D>::NI];
    kv_lane_offsets<D, false>(a.k_ts, wave, lane, kvo);
    kv_lane_offsets<D, true>(a.v_ts, wave, lane, vvo);

    // ---- Q fragments (B operand): lane (q = l31, hh) holds Q[q][16*ks + 8*hh .. +7] ----
    // rows of this wave: body rows qx + 32 g + l31; group a.cls_wave (> 0: the spare wave of the last block) is the class row alone
    const int grp = qt * 4 + wave;
    const bool cls_grp = a.cls_wave > 0 && grp == a.cls_wave;
    const int q_row = cls_grp ? (l31 == 0 ? 0 : a.S) : a.qx + grp * 32 + l31;
    const int q_ld = q_row < a.S ? q_row : a.S - 1;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8_t *>(qb + (long)q_ld * a.q_ts + ks * 16 + hh * 8);

    f32x16_t o[DB];
    float m_run = -1.0e30f, l_run = 0.f;
    const float c2 = a.scale_log2e;
    if (a.kx) {
        // token 0 as the initial state: s0 = q . k0 (this lane's half of the channels, then the other half's), p0 = 1, O = v0
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4_t kk = *reinterpret_cast<const uint4_t *>(k0_ + ks * 16 + hh * 8);
            const uint4_t qq = __builtin_bit_cast(uint4_t, qf[ks]);
            acc = dot2_acc<F16>(qq.x, kk.x, acc);
            acc = dot2_acc<F16>(qq.y, kk.y, acc);
            acc = dot2_acc<F16>(qq.z, kk.z, acc);
            acc = dot2_acc<F16>(qq.w, kk.w, acc);
        }
        m_run = halves_sum(acc) * c2;
        l_run = hh == 0 ? 1.f : 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const uint2_t vv = *reinterpret_cast<const uint2_t *>(v0_ + d * 32 + 8 * rq + 4 * hh);
                o[d][4 * rq] = cvt16<F16>(vv.x & 0xffffu);
                o[d][4 * rq + 1] = cvt16<F16>(vv.x >> 16);
                o[d][4 * rq + 2] = cvt16<F16>(vv.y & 0xffffu);
                o[d][4 * rq + 3] = cvt16<F16>(vv.y >> 16);
            }
    } else {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    }
    const int nkt = (Sk + KVBLK - 1) / KVBLK;

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
    // NKB = 1: the last tile holds <= 32 live keys -- second key block skipped
    auto qk = [&](uint32_t ks_, f32x16_t (&st)[2], auto nkb_) {
        constexpr int NKB = decltype(nkb_)::value;
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
    };

    // online softmax of one score tile (raw scores; scale*log2e folded into the exp2 argument) + O^T += V^T P^T.
    // Lane holds keys kb*32 + (r&3) + 8*(r>>2) + 4*hh of query l31.  Rescale of O / l is DEFERRED while the running
    // max grows by less than THR (exp2 domain): P is then bounded by 2^THR instead of 1 (fp32 accumulation).
    auto softmax_pv = [&](f32x16_t (&st)[2], uint32_t vs_, int k0, auto nkb_, auto mask_) {
        constexpr bool MASK = decltype(mask_)::value;   // false: the tile is known to be full (steady-state steps)
        constexpr int NKB = decltype(nkb_)::value;
        constexpr float THR = DEFER ? 6.0f : 0.0f;
        float mx = -1.0e30f;
        if (MASK && k0 + KVBLK > Sk) {   // tail tile: mask keys >= Sk (block-uniform branch)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    st[kb][r] = key < Sk ? st[kb][r] : -1.0e30f;
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
                    const int blk = (kb * 32 + 16 * u) * (D * 2);   // immediate; key2 = key1 + 8 shares the swizzle
                    const s16x4_t v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk));
                    const s16x4_t v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk + 8 * (D * 2)));
                    const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    o[d] = mfma16<F16>(vf, pf, o[d]);
                }
            }
    };

    constexpr std::integral_constant<int, 2> FULL{};
    constexpr std::integral_constant<int, 1> HALF{};
    {
        // K_t and V_t staged together one tile ahead; QK -> softmax -> PV in sequence.  Waves whose 32 query rows are all
        // padding only stage and synchronise.
        const bool live_wave = a.no_trim || cls_grp || a.qx + grp * 32 < a.S;
        const bool short_tail = !a.no_trim && Sk - (nkt - 1) * KVBLK <= 32;
        stage_kv<D, false>(kb_, a.k_ts, 0, Sk, smem, wave, lane, kvo);
        stage_kv<D, true>(vb_, a.v_ts, 0, Sk, smem + TILE, wave, lane, vvo);
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
                stage_kv<D, false, NEXT == 2>(kb_, a.k_ts, (t + 1) * KVBLK, Sk, nx, wave, lane, kvo);
                stage_kv<D, true, NEXT == 2>(vb_, a.v_ts, (t + 1) * KVBLK, Sk, nx + TILE, wave, lane, vvo);
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
        if (nkt >= 1) { if (short_tail) tile_step(nkt - 1, HALF, CDYN, C0); else tile_step(nkt - 1, FULL, CDYN, C0); }
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
            const int qr = cls_grp ? (row == 0 ? 0 : a.S) : a.qx + grp * 32 + row;
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

template <int D, bool DEFER, bool F16, int EPI>
static void attn_launch_one(const AttnArgs &a, unsigned grid, hipStream_t st)
{
    VLLM_LAUNCH((attn_fwd_kernel<D, DEFER, F16, EPI>), dim3(grid), dim3(ATT_THREADS), 4 * (size_t)KVBLK * D * 2, st, a);
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
    // attn_variant: 32 = automatic (deferred rescale, O through LDS, class-token split).  bit 1: deferred rescale; bit 4: no padding
    // trim (and no split); bit 6: O through LDS; bit 7: NO class-token split; bit 10: class token out of the KEY tiling only (what
    // automatic does anyway when the body rows leave no spare wave).  Bits 0, 2, 3 (rounds 2-5 schedules) are ignored.
    int var = attn_variant();
    if (var & 32) var = 2 | 64;
    a.no_trim = (var >> 4) & 1;
    const bool defer = (var >> 1) & 1;
    const bool epi_lds = ((var >> 6) & 1) && aligned16(a.out);
    // the class-token split (attn_common.hpp): S = 64 n + 1
    const bool split = !((var >> 7) & 1) && !a.no_trim && a.S > 64 && (a.S - 1) % KVBLK == 0;
    const bool spare_wave = ((a.S - 1) / 32) % 4 != 0;          // the body rows do not fill their last block
    a.kx = split ? 1 : 0;
    a.qx = split && spare_wave && !((var >> 10) & 1) ? 1 : 0;
    const int rows = a.S - a.qx;
    a.nqt = (rows + QBLK - 1) / QBLK;
    a.row0 = 0;
    a.cls_wave = a.qx ? rows / 32 : 0;
    const long groups = ((long)a.B * a.H + 7) / 8;
    const unsigned grid = (unsigned)(groups * 8 * a.nqt);
#define LD(DD) do { \
        if (a.f16) { \
            if (epi_lds) attn_launch_one<DD, true, true, 1>(a, grid, st); \
            else attn_launch_one<DD, true, true, 0>(a, grid, st); \
        } else if (!defer) attn_launch_one<DD, false, false, 0>(a, grid, st); \
        else if (epi_lds) attn_launch_one<DD, true, false, 1>(a, grid, st); \
        else attn_launch_one<DD, true, false, 0>(a, grid, st); \
    } while (0)
    if (D == 64) LD(64); else LD(128);
#undef LD
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
