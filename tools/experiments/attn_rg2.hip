// Moved out of visionllm_amd/csrc/attn.hip (round 2): measured slower than the one-group kernel, not part of the library.
// Needs the helpers of attn.hip (KvStage, kv_lane_offsets, stage_kv, halves_max / halves_sum) to build.
// ---------------------------------------------------------------------------------------------------------------------
// d = 64, TWO 32-row query groups per wave ("attn_variant" bit 6; opt-in).
// STATUS (round 1): correct (same tests), NOT the default: 86.5 us against 75.6 us for the one-group kernel on the same box.
// The two-group body needs 256 VGPRs (2 waves per SIMD instead of 4; bounding it to 168 spills 466 registers), and the
// occupancy it gives up costs more than the fragment reads it saves.
// The ablation table (profiles/r01_attn_ablation_and_clocks.txt) has the K/V fragment reads at 19 of 80 us on a kernel that
// runs at the package power limit: with 64 query rows per wave every ds_read_b128 of K and every transposed read of V feeds
// two MFMAs instead of one.  Block = 2 waves = the same 128 query rows (same grid, same LDS ring, same staging helpers with
// 2 waves sharing the DMA); plain schedule with deferred rescale, peeled short last tile, ring-parity unrolled steady state.
// Row group 1 of a wave whose rows are all padding (S = 577: the last wave of the last block) is skipped.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ATT2_THREADS = 128;

__global__ __launch_bounds__(ATT2_THREADS, 2) void attn_fwd_rg2_kernel(const AttnArgs a)
{
    constexpr int D = 64, KS = D / 16, DB = D / 32, RG = 2, WPB = 2;
    constexpr int TILE = KVBLK * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K | V]
    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int bh = (sidx / a.nqt) * 8 + xcd;
    const int qt = sidx % a.nqt;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, head = bh % a.H;
    const uint16_t *qb = a.q + (long)b * a.q_bs + (long)head * a.q_hs;
    const uint16_t *kb_ = a.k + (long)b * a.k_bs + (long)head * a.k_hs;
    const uint16_t *vb_ = a.v + (long)b * a.v_bs + (long)head * a.v_hs;

    uint32_t kvo[KvStage<D, WPB>::NI], vvo[KvStage<D, WPB>::NI];
    kv_lane_offsets<D, false, WPB>(a.k_ts, wave, lane, kvo);
    kv_lane_offsets<D, true, WPB>(a.v_ts, wave, lane, vvo);

    const int row0 = qt * QBLK + wave * (32 * RG);
    const bool live1 = a.no_trim || row0 + 32 < a.S;   // wave-uniform: the second row group holds a live query
    bf16x8_t qf[RG][KS];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int q_row = row0 + g * 32 + l31;
        const int q_ld = q_row < a.S ? q_row : a.S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[g][ks] = *reinterpret_cast<const bf16x8_t *>(qb + (long)q_ld * a.q_ts + ks * 16 + hh * 8);
    }
    f32x16_t o[RG][DB];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][d][r] = 0.f;
    float m_run[RG] = {-1.0e30f, -1.0e30f}, l_run[RG] = {0.f, 0.f};
    const float c2 = a.scale_log2e;
    const int nkt = (a.S + KVBLK - 1) / KVBLK;

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
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};

    // one K/V tile: NKB key blocks of 32; MASK: keys >= S are masked (ragged last tile only)
    auto tile_body = [&](uint32_t ks_, uint32_t vs_, int k0, auto nkb_, auto mask_) __attribute__((always_inline)) {
        constexpr int NKB = decltype(nkb_)::value;
        constexpr bool MASK = decltype(mask_)::value;
        constexpr float THR = 6.0f;
        f32x16_t st[RG][2];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[g][kb][r] = 0.f;
        // S^T = K Q^T: every K fragment feeds both row groups
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *(const __attribute__((address_space(3))) bf16x8_t *)(uintptr_t)(ks_ + kofs[ks] + kb * 32 * (D * 2));
                st[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], st[0][kb], 0, 0, 0);
                if (live1) st[1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], st[1][kb], 0, 0, 0);
            }
        // online softmax per row group (lane holds keys kb*32 + (r&3) + 8*(r>>2) + 4*hh of query l31 of the group)
        uint32_t pk[RG][2][8];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            if (g == 1 && !live1) break;
            float mx = -1.0e30f;
            if (MASK && k0 + KVBLK > a.S) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        st[g][kb][r] = key < a.S ? st[g][kb][r] : -1.0e30f;
                    }
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[g][kb][r]);
            mx = halves_max(mx) * c2;
            if (!__all(mx - m_run[g] <= THR)) {
                const float m_new = fmaxf(m_run[g], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
                m_run[g] = m_new;
                l_run[g] *= alpha;
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[g][d][r] *= alpha;
            }
            float psum[2] = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(st[g][kb][r], c2, -m_run[g]));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(st[g][kb][r + 1], c2, -m_run[g]));
                    const uint32_t w = pack_bf16x2(p0, p1);
                    pk[g][kb][r >> 1] = w;
                    psum[(r >> 1) & 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, psum[(r >> 1) & 1], false);
                }
            l_run[g] += psum[0] + psum[1];
        }
        // O^T += V^T P^T: every transposed V fragment feeds both row groups
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 pw0 = {pk[0][kb][4 * u], pk[0][kb][4 * u + 1], pk[0][kb][4 * u + 2], pk[0][kb][4 * u + 3]};
                const u32x4 pw1 = {pk[1][kb][4 * u], pk[1][kb][4 * u + 1], pk[1][kb][4 * u + 2], pk[1][kb][4 * u + 3]};
                const bf16x8_t pf0 = __builtin_bit_cast(bf16x8_t, pw0), pf1 = __builtin_bit_cast(bf16x8_t, pw1);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int blk = (kb * 32 + 16 * u) * (D * 2);
                    const s16x4_t v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk));
                    const s16x4_t v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk + 8 * (D * 2)));
                    const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    o[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf0, o[0][d], 0, 0, 0);
                    if (live1) o[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf1, o[1][d], 0, 0, 0);
                }
            }
    };

    constexpr std::integral_constant<int, 2> FULL{};
    constexpr std::integral_constant<int, 1> HALF{};
    constexpr std::integral_constant<int, 0> C0{};
    constexpr std::integral_constant<int, 1> C1{};
    constexpr std::integral_constant<int, 2> C2{};
    constexpr std::integral_constant<int, -1> CDYN{};
    const bool short_tail = !a.no_trim && a.S - (nkt - 1) * KVBLK <= 32;
    stage_kv<D, false, true, WPB>(kb_, a.k_ts, 0, a.S, smem, wave, lane, kvo);
    stage_kv<D, true, true, WPB>(vb_, a.v_ts, 0, a.S, smem + TILE, wave, lane, vvo);
    auto tile_step = [&](int t, auto nkb_, auto stage_, auto next_) __attribute__((always_inline)) {
        constexpr int STAGE = decltype(stage_)::value, NEXT = decltype(next_)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int slot = STAGE >= 0 ? STAGE : (t & 1);
        const uint32_t ks_ = (uint32_t)(slot * 2 * TILE);
        if constexpr (NEXT != 0) {
            char *nx = smem + (slot ^ 1) * 2 * TILE;
            stage_kv<D, false, NEXT == 2, WPB>(kb_, a.k_ts, (t + 1) * KVBLK, a.S, nx, wave, lane, kvo);
            stage_kv<D, true, NEXT == 2, WPB>(vb_, a.v_ts, (t + 1) * KVBLK, a.S, nx + TILE, wave, lane, vvo);
        }
        tile_body(ks_, ks_ + TILE, t * KVBLK, nkb_, std::integral_constant<bool, NEXT != 1>{});
    };
    const int n_main = nkt - 2;
    for (int t = 0; t < n_main; t += 2) {
        tile_step(t, FULL, C0, C1);
        if (t + 1 < n_main) tile_step(t + 1, FULL, C1, C1);
    }
    if (nkt >= 2) tile_step(nkt - 2, FULL, CDYN, C2);
    if (short_tail) tile_step(nkt - 1, HALF, CDYN, C0); else tile_step(nkt - 1, FULL, CDYN, C0);

#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int q_row = row0 + g * 32 + l31;
        const float l_tot = halves_sum(l_run[g]);
        const float inv = 1.0f / l_tot;
        if (q_row < a.S) {
            uint16_t *orow = a.out + (((long)b * a.S + q_row) * a.H + head) * D;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    uint2_t w;
                    w.x = pack_bf16x2(o[g][d][4 * rq] * inv, o[g][d][4 * rq + 1] * inv);
                    w.y = pack_bf16x2(o[g][d][4 * rq + 2] * inv, o[g][d][4 * rq + 3] * inv);
                    *reinterpret_cast<uint2_t *>(orow + d * 32 + 8 * rq + 4 * hh) = w;
                }
        }
    }
}

