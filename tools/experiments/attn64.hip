// NOT part of the library since the end of round 6.  It was built in three steps, tested (95-109 attention tests green with it forced
// on) and measured on MI355X at 40 x 25 x S1025 x d128 against attn.hip's 641-650 us:
//   v0  whole-tile software pipeline (two generations of 64-key scores): 640 registers as the compiler allocates them, 128 spilled;
//   v1  half-tile pipeline, compiler-placed fragment reads: 854 us (d 64, 40 x 16 x S577: 175 us against 78);
//   v2  (this file) K fragments by inline assembly in two register sets with counted waits, one rescale branch per half tile at a
//       point with no read in flight, two max chains: 883 us -- the fragment latency was not what the time is made of.
// profiles/r06_attn64.txt: timings, the issue-gap histogram of the ISA and the phase clock of v2 (s_memtime, every wave): of a
// wave's 130 K cycles per 16-tile block 15.7 % are prologue (Q, first tiles' DMA, first scores: nothing else runs on the SIMD),
// 14.5 % the per-tile vmcnt(0) + block barrier (skew between four waves that no other wave covers), 4.4 % the epilogue, and the four
// compute phases take 4.97 K cycles per 64-key tile against 2.05 K of MFMA pipe time -- 8.5 non-MFMA issues per MFMA where the
// guide budgets 5.  What it would take: a persistent block loop with K / V / Q prefetched across block seams (the 20 % of
// prologue + epilogue), hand-allocated AGPRs (the compiler keeps the score accumulators in AGPRs and copies them out: 32-44
// v_accvgpr_read per half tile) and hand-placed fillers -- a hand-scheduled instruction stream.
// To rebuild: copy next to attn.hip, declare `int cls_block` in AttnArgs, call attn_fwd64_launch from attn_fwd_launch (attn_variant
// bits 11-12 = 2 in round 6), and route vllm_debug_counters to attn64_debug_counters for the phase clock (bit 13).
//
// Fused self-attention forward, the 64-rows-per-wave body (round 6).
//
// Same operator, layouts and arithmetic as attn.hip (flash_attention.py:30-75 / modeling_intern_vit.py:136-140): S^T = K Q^T on
// v_mfma_f32_32x32x16, fp32 online softmax in the exp2 domain with deferred rescale, P rounded to 16 bits, O^T += V^T P^T with V
// read through ds_read_b64_tr_b16.  What differs is the shape of a wave:
//   * a wave owns 64 query rows = TWO 32-row groups, and every K / V fragment it reads from LDS feeds two MFMAs (one per group):
//     attn.hip's 32-row waves read the whole K and V tile per 32 rows -- at 16 resident waves that is 256 KB of LDS reads per
//     64-key step of a CU against 2048 MFMA cycles: the LDS pipe is as long as the matrix pipe.  Here it is half of it;
//   * one wave per SIMD (4 waves = 256 query rows per block, one block per CU) with the whole register file: O (2 x D / 32
//     accumulators), two generations of scores, Q of both groups;
//   * software pipeline inside the wave at HALF-tile (32-key) granularity, K running ahead of V: the scores of half tile h + 1
//     (MFMA) are issued next to the exponentials of half tile h (VALU), then P V of h (MFMA) next to the row maxima of h + 1
//     (VALU).  No other wave shares the SIMD, so the two instruction streams of ONE wave have to fill each other's gaps.  (Whole-tile
//     granularity -- two generations of 64-key scores -- needs 640 registers as the compiler allocates them: 128 spilled);
//   * K / V by LDS-DMA into a ring of three K and two V slots, one block barrier per 64-key tile.
// The key tiling must be exact ((S - kx) % 64 == 0: the class-token split of attn_common.hpp, or S % 64 == 0); ragged query rows
// are clamped on load and dropped on store.  The launcher (attn.hip) decides which body runs.
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "attn_common.hpp"

namespace vllm {

__device__ __forceinline__ float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Query row 0 of one (tile, head) on the VALU, by ONE BLOCK of NW waves.  Scores: D / 8 lanes per key row (one coalesced 16-byte
// chunk each), 64 / (D / 8) key rows per wave instruction, wave w takes every NW-th group of rows; the softmax goes through `ps`
// (S floats of LDS); P V: the same lane <-> (key phase, chunk) mapping, partial sums combined over the key phases by lane exchanges
// and over the waves through LDS.  ~2 S D multiply-adds: 1-4 us per block, hidden among the MFMA blocks it is interleaved with.
template <int D, int NW, bool F16>
__device__ __forceinline__ void attn_class_row(const AttnArgs &a, int bh, int wave, int lane, char *smem)
{
    constexpr int CPR = D / 8, KPI = 64 / CPR;
    const int S = a.S;
    float *ps = reinterpret_cast<float *>(smem);
    float *red = ps + ((S + 3) & ~3);          // [NW] maxima, [NW] sums, [NW][D] partial outputs
    const int b = bh / a.H, head = bh % a.H;
    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *kb_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *vb_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;
    const int c = lane % CPR, kq = lane / CPR;
    const uint4_t qv = *reinterpret_cast<const uint4_t *>(qb + c * 8);
    float mx = -1.0e30f;
#pragma unroll 4
    for (int j = wave * KPI + kq; j < S; j += NW * KPI) {
        const uint4_t kk = *reinterpret_cast<const uint4_t *>(kb_ + (long)j * a.k_ts + c * 8);
        float acc = dot2_acc<F16>(qv.x, kk.x, 0.f);
        acc = dot2_acc<F16>(qv.y, kk.y, acc);
        acc = dot2_acc<F16>(qv.z, kk.z, acc);
        acc = dot2_acc<F16>(qv.w, kk.w, acc);
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
        const float sc = acc * a.scale_log2e;
        if (c == 0) ps[j] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int j = wave * 64 + lane; j < S; j += NW * 64) {   // P rounded to 16 bits like the MFMA path's operand
        const float pr = cvt16<F16>(pack16x2<F16>(__builtin_amdgcn_exp2f(ps[j] - mx), 0.f) & 0xffffu);
        ps[j] = pr;
        sum += pr;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[NW + wave] = sum;
    __syncthreads();
    sum = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[NW + w];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int j = wave * KPI + kq; j < S; j += NW * KPI) {
        const float pj = ps[j];
        const uint4_t vv = *reinterpret_cast<const uint4_t *>(vb_ + (long)j * a.v_ts + c * 8);
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = fmaf(pj, cvt16<F16>(w[i] & 0xffffu), acc[2 * i]);
            acc[2 * i + 1] = fmaf(pj, cvt16<F16>(w[i] >> 16), acc[2 * i + 1]);
        }
    }
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    float *po = red + 2 * NW;
    if (kq == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) po[wave * D + c * 8 + i] = acc[i];
    }
    __syncthreads();
    if (wave == 0 && lane < CPR) {
        const float inv = 1.0f / sum;
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[i] = po[lane * 8 + i];
#pragma unroll
            for (int w = 1; w < NW; ++w) r[i] += po[w * D + lane * 8 + i];
        }
        uint16_t *orow = a.out + (((long)b * S) * a.H + head) * D + lane * 8;
        *reinterpret_cast<uint2_t *>(orow) = uint2_t{pack16x2<F16>(r[0] * inv, r[1] * inv), pack16x2<F16>(r[2] * inv, r[3] * inv)};
        *reinterpret_cast<uint2_t *>(orow + 4) = uint2_t{pack16x2<F16>(r[4] * inv, r[5] * inv), pack16x2<F16>(r[6] * inv, r[7] * inv)};
    }
}


__device__ unsigned long long g_a64_prof[8];   // phase clock (attn_variant bit 13): ticks of every wave, summed

template <int D, bool F16, bool PROF = false>
__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(const AttnArgs a)
{
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto tick = [&](int slot) {
        if constexpr (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            pacc[slot] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (PROF) tprev = __builtin_amdgcn_s_memtime();
    constexpr int KS = D / 16;            // k-steps of the QK^T product
    constexpr int DB = D / 32;            // 32-wide output blocks
    constexpr int TILE = KVBLK * D * 2;   // bytes per K or V tile
    constexpr float THR = 6.0f;           // deferred rescale: P <= 2^THR
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K slot 0 | K slot 1 | K slot 2 | V slot 0 | V slot 1]

    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();   // fragment reads address LDS by byte offset: no static LDS here
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- block -> (b, head, q tile): all q tiles of a (b, head) on one XCD; with a.cls_block the pair's last "tile" is its class row ----
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int ntile = a.nqt + a.cls_block;
    const int bh = (sidx / ntile) * 8 + xcd;
    const int qt = sidx % ntile;
    if (bh >= a.B * a.H) return;
    if (qt == a.nqt) {
        attn_class_row<D, 4, F16>(a, bh, wave, lane, smem);
        return;
    }
    const int b = bh / a.H, head = bh % a.H;
    const int Sk = a.S - a.kx;

    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *k0_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *v0_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;
    const uint16_t *kb_ = k0_ + (long)a.kx * a.k_ts;
    const uint16_t *vb_ = v0_ + (long)a.kx * a.v_ts;

    uint32_t kvo[KvStage<D>::NI], vvo[KvStage<D>::NI];
    kv_lane_offsets<D, false>(a.k_ts, wave, lane, kvo);
    kv_lane_offsets<D, true>(a.v_ts, wave, lane, vvo);

    // ---- Q fragments (B operand) of both groups: lane (q = l31, hh) holds Q[q][16*ks + 8*hh .. +7] ----
    const int row_w = a.qx + qt * 256 + wave * 64;      // first query row of this wave
    const bool live = row_w < a.S;
    bf16x8_t qf[2][KS];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int q_row = row_w + g * 32 + l31;
        const int q_ld = q_row < a.S ? q_row : a.S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[g][ks] = *reinterpret_cast<const bf16x8_t *>(qb + (long)q_ld * a.q_ts + ks * 16 + hh * 8);
    }

    f32x16_t o[2][DB];
    float m_run[2] = {-1.0e30f, -1.0e30f}, l_run[2] = {0.f, 0.f};
    const float c2 = a.scale_log2e;
    if (a.kx) {
        // token 0 as the initial state: s0 = q . k0 (this lane's half of the channels, then the other half's), p0 = 1, O = v0
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float acc = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4_t kk = *reinterpret_cast<const uint4_t *>(k0_ + ks * 16 + hh * 8);
                const uint4_t qq = __builtin_bit_cast(uint4_t, qf[g][ks]);
                acc = dot2_acc<F16>(qq.x, kk.x, acc);
                acc = dot2_acc<F16>(qq.y, kk.y, acc);
                acc = dot2_acc<F16>(qq.z, kk.z, acc);
                acc = dot2_acc<F16>(qq.w, kk.w, acc);
            }
            m_run[g] = halves_sum(acc) * c2;
            l_run[g] = hh == 0 ? 1.f : 0.f;
        }
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const uint2_t vv = *reinterpret_cast<const uint2_t *>(v0_ + d * 32 + 8 * rq + 4 * hh);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    o[g][d][4 * rq] = cvt16<F16>(vv.x & 0xffffu);
                    o[g][d][4 * rq + 1] = cvt16<F16>(vv.x >> 16);
                    o[g][d][4 * rq + 2] = cvt16<F16>(vv.y & 0xffffu);
                    o[g][d][4 * rq + 3] = cvt16<F16>(vv.y >> 16);
                }
            }
    } else {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[g][d][r] = 0.f;
    }
    const int nkt = Sk / KVBLK;           // exact (launcher)

    // loop-invariant LDS offsets of the fragments (attn.hip)
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

    typedef f32x16_t scores_t[2];         // [group]: the scores of one 32-key half tile
    uint32_t pk[2][8];                    // P of the current half tile, packed pairs: [group][register pair]

    // K fragments by inline assembly, four at a time into one of two register sets, with counted waits tied to the registers they
    // release (the compiler serialises ds_read_b128 -> s_waitcnt lgkmcnt(0) -> MFMA through ONE register quad: with nothing else on the
    // SIMD every fragment's LDS latency is then exposed).  Rules inherited from msda_tiled9.hip: no control flow between a request and
    // its wait (a join may copy registers whose reads are still in flight), no "memory" clobber on the requests.
    struct KSet { bf16x8_t f0, f1, f2, f3; };
    auto req4 = [&](KSet &s_, uint32_t base, int ks0) {
        const uint32_t a0 = base + kofs[ks0], a1 = base + kofs[ks0 + 1], a2 = base + kofs[ks0 + 2], a3 = base + kofs[ks0 + 3];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                     : "=&v"(s_.f0), "=&v"(s_.f1), "=&v"(s_.f2), "=&v"(s_.f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    };
    auto wait4 = [&](KSet &s_, auto cnt_) {
        constexpr int CNT = decltype(cnt_)::value;
        if constexpr (CNT == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(s_.f0), "+v"(s_.f1), "+v"(s_.f2), "+v"(s_.f3));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s_.f0), "+v"(s_.f1), "+v"(s_.f2), "+v"(s_.f3));
    };
    constexpr std::integral_constant<int, 4> W4{};
    constexpr std::integral_constant<int, 0> W0{};
    // S^T of one 32-key half of a K tile for both groups: every K fragment feeds two MFMAs; the eight fragments are requested up
    // front (the exponentials of the previous half tile are issued under their flight)
    KSet ka, kb2;
    auto qk_req = [&](uint32_t ks_, int kb) {
        const uint32_t base = ks_ + kb * 32 * (D * 2);
        req4(ka, base, 0);
        req4(kb2, base, 4);
    };
    auto qk_mma = [&](scores_t &st) {
        static_assert(KS == 8 || KS == 4, "two sets of four fragments (d 128) or one (d 64: the second set repeats the first)");
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        wait4(ka, W4);
        st[0] = mfma16<F16>(ka.f0, qf[0][0], zero);
        st[1] = mfma16<F16>(ka.f0, qf[1][0], zero);
        st[0] = mfma16<F16>(ka.f1, qf[0][1], st[0]);
        st[1] = mfma16<F16>(ka.f1, qf[1][1], st[1]);
        st[0] = mfma16<F16>(ka.f2, qf[0][2], st[0]);
        st[1] = mfma16<F16>(ka.f2, qf[1][2], st[1]);
        st[0] = mfma16<F16>(ka.f3, qf[0][3], st[0]);
        st[1] = mfma16<F16>(ka.f3, qf[1][3], st[1]);
        wait4(kb2, W0);
        if constexpr (KS == 8) {
            st[0] = mfma16<F16>(kb2.f0, qf[0][4], st[0]);
            st[1] = mfma16<F16>(kb2.f0, qf[1][4], st[1]);
            st[0] = mfma16<F16>(kb2.f1, qf[0][5], st[0]);
            st[1] = mfma16<F16>(kb2.f1, qf[1][5], st[1]);
            st[0] = mfma16<F16>(kb2.f2, qf[0][6], st[0]);
            st[1] = mfma16<F16>(kb2.f2, qf[1][6], st[1]);
            st[0] = mfma16<F16>(kb2.f3, qf[0][7], st[0]);
            st[1] = mfma16<F16>(kb2.f3, qf[1][7], st[1]);
        }
    };
    // row maxima of a half tile (both groups): VALU only, issued next to the P V MFMAs
    auto row_max = [&](scores_t &st, float (&mx)[2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float m0 = fmaxf(st[g][0], st[g][1]), m1 = fmaxf(st[g][2], st[g][3]);   // (two chains: half the dependent latency)
#pragma unroll
            for (int r = 4; r < 16; r += 4) {
                m0 = fmaxf(m0, fmaxf(st[g][r], st[g][r + 1]));
                m1 = fmaxf(m1, fmaxf(st[g][r + 2], st[g][r + 3]));
            }
            mx[g] = halves_max(fmaxf(m0, m1)) * c2;
        }
    };
    // ONE wave-uniform branch for the running-max update, at a point where no LDS read is in flight: with the deferred rescale it is
    // rare after the first tiles
    auto rescale = [&](const float (&mx)[2]) {
        if (__builtin_expect(!__all(mx[0] - m_run[0] <= THR && mx[1] - m_run[1] <= THR), 0)) {
            // (the empty volatile statement keeps the compiler from flattening this branch: it had turned the rescale of the 2 x D / 32
            //  accumulators into unconditional code -- 128 AGPR reads, 64 packed multiplies, 128 AGPR writes per half tile)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float m_new = fmaxf(m_run[g], mx[g]);
                const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
                m_run[g] = m_new;
                l_run[g] *= alpha;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[g][d][r] *= alpha;
            }
        }
    };
    // exponentials of a half tile -> packed P, row sums from the ROUNDED probabilities (attn.hip)
    auto finish = [&](scores_t &st) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float psum[2] = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(st[g][r], c2, -m_run[g]));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(st[g][r + 1], c2, -m_run[g]));
                const uint32_t w = pack16x2<F16>(p0, p1);
                pk[g][r >> 1] = w;
                psum[(r >> 1) & 1] = dot2_ones<F16>(w, psum[(r >> 1) & 1]);
            }
            l_run[g] += psum[0] + psum[1];
        }
    };
    // O^T += V^T P^T of one 32-key half of a V tile for both groups: every V fragment feeds two MFMAs
    auto pv = [&](uint32_t vs_, int kb) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 pw0 = {pk[0][4 * u], pk[0][4 * u + 1], pk[0][4 * u + 2], pk[0][4 * u + 3]};
            const u32x4 pw1 = {pk[1][4 * u], pk[1][4 * u + 1], pk[1][4 * u + 2], pk[1][4 * u + 3]};
            const bf16x8_t pf0 = __builtin_bit_cast(bf16x8_t, pw0), pf1 = __builtin_bit_cast(bf16x8_t, pw1);
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int blk = (kb * 32 + 16 * u) * (D * 2);
                const s16x4_t v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk));
                const s16x4_t v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk + 8 * (D * 2)));
                const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                o[0][d] = mfma16<F16>(vf, pf0, o[0][d]);
                o[1][d] = mfma16<F16>(vf, pf1, o[1][d]);
            }
        }
    };

    // ring: three K slots, two V slots (K runs a tile and a half ahead of V: slot (t + 2) % 3 is free once every wave has left
    // the first half of tile t - 1 -- one block barrier per tile covers both refills)
    auto kslot = [&](int t) { return smem + (t % 3) * TILE; };
    auto vslot = [&](int t) { return smem + 3 * TILE + (t & 1) * TILE; };
    auto kaddr = [&](int t) { return (uint32_t)((t % 3) * TILE); };
    auto vaddr = [&](int t) { return (uint32_t)(3 * TILE + (t & 1) * TILE); };

    // prologue: K(0), V(0), K(1) landed; scores and maxima of the first half tile
    stage_kv<D, false, false>(kb_, a.k_ts, 0, Sk, kslot(0), wave, lane, kvo);
    stage_kv<D, true, false>(vb_, a.v_ts, 0, Sk, vslot(0), wave, lane, vvo);
    if (nkt > 1) stage_kv<D, false, false>(kb_, a.k_ts, KVBLK, Sk, kslot(1), wave, lane, kvo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    scores_t sA, sB;
    float mxA[2] = {0.f, 0.f}, mxB[2] = {0.f, 0.f};
    if (live) {
        qk_req(kaddr(0), 0);
        qk_mma(sA);
        row_max(sA, mxA);
    }
    tick(0);
    // tile t, first half:  [rescale for (t,0)] K fragments of (t,1) requested | exponentials of (t,0) | scores of (t,1) | P V of (t,0) next
    // to the maxima of (t,1);  second half: the same one half tile later, the scores being those of tile t + 1's first half
    auto tile = [&](int t, auto last_) {
        constexpr bool LAST = decltype(last_)::value;
        // K(t+1) and V(t) (issued one tile ago) have landed for every wave; every wave is done with K(t-1) and V(t-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tick(1);
        if (t + 2 < nkt) stage_kv<D, false, false>(kb_, a.k_ts, (t + 2) * KVBLK, Sk, kslot(t + 2), wave, lane, kvo);
        if (!LAST) stage_kv<D, true, false>(vb_, a.v_ts, (t + 1) * KVBLK, Sk, vslot(t + 1), wave, lane, vvo);
        tick(2);
        if (live) {
            rescale(mxA);
            qk_req(kaddr(t), 1);
            finish(sA);
            qk_mma(sB);
            tick(3);
            pv(vaddr(t), 0);
            row_max(sB, mxB);
            tick(4);
            rescale(mxB);
            if (!LAST) qk_req(kaddr(t + 1), 0);
            finish(sB);
            if (!LAST) qk_mma(sA);
            tick(5);
            pv(vaddr(t), 1);
            if (!LAST) row_max(sA, mxA);
            tick(6);
        }
    };
    for (int t = 0; t + 1 < nkt; ++t) tile(t, std::false_type{});
    tile(nkt - 1, std::true_type{});

    // ---- finalize: O / l through LDS, whole-row stores (attn.hip, EPI 1); a wave owns 64 rows x D x 2 bytes of the dead ring ----
    constexpr int CPRO = D / 8;
    static_assert(4 * 64 * D * 2 <= 5 * TILE, "the O tiles of a block must fit its K/V ring");
    __syncthreads();
    char *wbase = smem + wave * (64 * D * 2);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float inv = 1.0f / halves_sum(l_run[g]);
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2_t w;
                w.x = pack16x2<F16>(o[g][d][4 * rq] * inv, o[g][d][4 * rq + 1] * inv);
                w.y = pack16x2<F16>(o[g][d][4 * rq + 2] * inv, o[g][d][4 * rq + 3] * inv);
                const int chunk = d * 4 + rq;
                const int row = g * 32 + l31;
                *reinterpret_cast<uint2_t *>(wbase + row * (D * 2) + ((chunk ^ (row & (CPRO - 1))) << 4) + hh * 8) = w;
            }
    }
    constexpr int RPI = 64 / CPRO;
    const int rr = lane / CPRO, cc = lane % CPRO;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
        const int row = i * RPI + rr;
        const uint4_t v = *reinterpret_cast<const uint4_t *>(wbase + row * (D * 2) + ((cc ^ (row & (CPRO - 1))) << 4));
        const int qr = row_w + row;
        if (qr < a.S) *reinterpret_cast<uint4_t *>(a.out + (((long)b * a.S + qr) * a.H + head) * D + cc * 8) = v;
    }
    if constexpr (PROF) {
        tick(7);
        if (lane == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_a64_prof[i], pacc[i]);
    }
}

// a: kx / qx / cls_block set by the caller (attn_fwd_launch); requires (S - kx) % 64 == 0 and a 16-byte aligned output
int attn_fwd64_launch(AttnArgs a, int D, hipStream_t st)
{
    const int rows = a.S - a.qx;
    a.nqt = (rows + 255) / 256;
    a.cls_wave = 0;
    const long groups = ((long)a.B * a.H + 7) / 8;
    const dim3 grid((unsigned)(groups * 8 * (a.nqt + a.cls_block))), block(256);
    const size_t lds = 5 * (size_t)KVBLK * D * 2;
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd64_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd64_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
    }
    if (attn_variant() & 8192) {      // phase clock build (bf16, d 128)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd64_kernel<128, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
        if (D == 128 && !a.f16) {
            VLLM_LAUNCH((attn_fwd64_kernel<128, false, true>), grid, block, lds, st, a);
            VLLM_CHECK_LAUNCH("attn_fwd64_kernel<prof>");
            return VLLM_OK;
        }
    }
    if (D == 64) {
        if (a.f16) VLLM_LAUNCH((attn_fwd64_kernel<64, true>), grid, block, lds, st, a);
        else VLLM_LAUNCH((attn_fwd64_kernel<64, false>), grid, block, lds, st, a);
    } else {
        if (a.f16) VLLM_LAUNCH((attn_fwd64_kernel<128, true>), grid, block, lds, st, a);
        else VLLM_LAUNCH((attn_fwd64_kernel<128, false>), grid, block, lds, st, a);
    }
    VLLM_CHECK_LAUNCH("attn_fwd64_kernel");
    return VLLM_OK;
}

int attn64_debug_counters(long *out, int n)
{
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_a64_prof), sizeof(h)) != hipSuccess) return VLLM_ELAUNCH;
    for (int i = 0; i < n && i < 8; ++i) out[i] = (long)h[i];
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a64_prof), z, sizeof(z));
    return VLLM_OK;
}

}  // namespace vllm
