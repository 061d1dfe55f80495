// Round-2 experiment, moved out of visionllm_amd/csrc/attn.hip: NOT faster than the plain schedule, not part of the library.
//
// Idea: the softmax's per-score multiply-add and the per-tile row maximum disappear -- Q is pre-multiplied by
// scale * log2(e), the running reference maximum is subtracted BY THE MFMA (one extra k-step per 32-key block with the
// constant A operand (1, 0, ..., 0) and -m in the B operand), the maximum is only moved when a tile's row sum explodes.
// Steady state per 64-key tile and lane: 32 v_exp + 16 v_cvt_pk + 16 v_dot2c + ~8 (against ~165 VALU) and 18 MFMAs (16).
// A second template parameter runs 8-wave blocks (256 query rows: 3/5 of the K/V staging volume at S = 577).
//
// Measured on MI355X (tools/bench_attn.py, tools/dbg_attn3.py; gpurun_out/r02k_attn.txt, r02l_attn.txt), ViT-L 40 x 16 x S577 x d64:
//     plain schedule 78.5 us | folded, 4-wave blocks 77.8-78.6 us | folded, 8-wave blocks + 4-wave remainder launch 87.8 us
//     on ALL-ZERO inputs: 66.0 | 66.2 | 80.0 us      (S1025 d128, 40 tiles: 650 | 641 | 718 us; zeros 545 | 554 | 656)
// i.e. 35 % fewer VALU instructions buy nothing, and the same instruction stream runs 20 % faster on zero data: the kernel
// is clock / power managed (DVFS), time follows energy, and two extra MFMAs cost about what ~70 VALU instructions save.
// The folded kernel also rounds q * scale * log2(e) to bf16 (per-element bound 1 ulp + 2^-7 sum p|v| instead of 2^-8).
// Needs the helpers of attn.hip (KvStage, kv_lane_offsets, stage_kv, halves_max / halves_sum) and AttnArgs::row0 to build;
// the launcher branch it used is at the end of this file.

// ---------------------------------------------------------------------------------------------------------------------
// "Folded" schedule (default since round 2; "attn_variant" 64 | ...): the softmax's per-score multiply-add disappears.
//
// The PMC profile of the plain schedule (profiles/r01_pmc_attn.txt) has 11 VALU per MFMA and the two pipes do not overlap
// inside a SIMD (an MFMA leaves ~9 cycles of VALU issue to the other waves): at d = 64 a 64-key tile is 512 cycles of MFMA
// next to ~750 cycles of VALU (32 exp, 32 fma, 16 max3, 16 cvt, 16 dot2, ...), so the matrix pipe cannot be busy more than
// a third of the time.  This variant moves work from the VALU to the matrix pipe and drops the per-tile maximum:
//   * Q is pre-multiplied by scale * log2(e) once per block (rounded to bf16 -- the reference's own naive path rounds
//     q * scale to bf16 too, modeling_intern_vit.py:137);
//   * the running reference maximum m is SUBTRACTED BY THE MFMA: one extra k-step per 32-key block whose A operand is the
//     constant column (1, 0, ..., 0) -- no LDS read -- and whose B operand carries -m of the lane's query row, so the
//     accumulator comes out as s - m and the probability is ONE v_exp_f32 away (2 MFMAs = 64 matrix cycles replace 32
//     v_fma);  m only has to be the SAME number for every tile of a row, so it is kept as a bf16 value;
//   * no per-tile row maximum: m is set from the true maximum of the FIRST tile (so the row sum can never underflow) and
//     afterwards only moved when a tile's local row sum exceeds 2^12 (a score more than ~12 above m, or an overflow to
//     inf): that tile is then redone from its still-live scores with the true maximum, O and l are rescaled, -m is
//     updated.  On softmax inputs this is rare; correctness never depends on it being rare (tests force it).
// Steady state per 64-key tile and lane: 32 v_exp, 16 v_cvt_pk, 16 v_dot2c, ~8 more -- against ~165 before.
// ---------------------------------------------------------------------------------------------------------------------
// NWB waves per block = NWB * 32 query rows of one (tile, head).  A block streams the whole K and V of its head through
// the LDS-DMA path, and that path -- ~25 GB/s per CU, 6.2 TB/s chip-wide, the same ceiling the MSDA window staging runs into
// -- is what bounded the 4-wave kernel (486 MB staged per ViT-L launch = 78 us, whatever the arithmetic did): 8-wave blocks
// stage 3/5 (S = 577) or 5/9 (S = 1025) of that; the rows behind the last full 256-row block go to a 4-wave launch.
template <int D, int NWB>
__global__ __launch_bounds__(NWB * 64, D == 64 ? 4 : 2) void attn_fwd_fold_kernel(const AttnArgs a)
{
    constexpr int KS = D / 16, DB = D / 32, TILE = KVBLK * D * 2, QB = NWB * 32;
    constexpr float TRIG = 4096.0f;       // lane-local row sum of one tile (32 keys) that forces a re-reference
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
    uint32_t kvo[KvStage<D, NWB>::NI], vvo[KvStage<D, NWB>::NI];
    kv_lane_offsets<D, false, NWB>(a.k_ts, wave, lane, kvo);
    kv_lane_offsets<D, true, NWB>(a.v_ts, wave, lane, vvo);

    // ---- Q fragments (B operand), pre-scaled: lane (q = l31, hh) holds Q[q][16*ks + 8*hh .. +7] * scale * log2(e) ----
    const int q_row = a.row0 + qt * QB + wave * 32 + l31;
    const int q_ld = q_row < a.S ? q_row : a.S - 1;
    const float c2 = a.scale_log2e;
    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint4_t raw = *reinterpret_cast<const uint4_t *>(qb + (long)q_ld * a.q_ts + ks * 16 + hh * 8);
        uint4_t sc;
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = pack_bf16x2(bf16lo_to_f32(raw[j]) * c2, bf16hi_to_f32(raw[j]) * c2);
        qf[ks] = __builtin_bit_cast(bf16x8_t, sc);
    }
    // the extra k-step: A = (1, 0, ..., 0) for every key (lanes hh == 0 hold k-slots 0..7), B = (-m, 0, ..., 0)
    const bf16x8_t kx = {(short)(hh == 0 ? 0x3f80 : 0), 0, 0, 0, 0, 0, 0, 0};
    bf16x8_t qx = {0, 0, 0, 0, 0, 0, 0, 0};
    float m_ref = 0.f;                     // always a bf16 value

    f32x16_t o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float l_run = 0.f;
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

    // one K/V tile.  NKB key blocks of 32; MASK: keys >= S are masked (ragged last tile); FIRST: the tile that sets m
    auto tile_body = [&](uint32_t ks_, uint32_t vs_, int k0, auto nkb_, auto mask_, auto first_) __attribute__((always_inline)) {
        constexpr int NKB = decltype(nkb_)::value;
        constexpr bool MASK = decltype(mask_)::value, FIRST = decltype(first_)::value;
        f32x16_t st[2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kf = *(const __attribute__((address_space(3))) bf16x8_t *)(uintptr_t)(ks_ + kofs[ks] + kb * 32 * (D * 2));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
            }
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx, qx, st[kb], 0, 0, 0);   // ... - m
        }
        if (MASK && k0 + KVBLK > a.S) {   // tail tile: mask keys >= S (block-uniform branch)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    st[kb][r] = key < a.S ? st[kb][r] : -1.0e30f;
                }
        }
        float psum[2] = {0.f, 0.f};
        uint32_t pk[2][8];
        bool redo = FIRST;
        if (!FIRST) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t w = pack_bf16x2(__builtin_amdgcn_exp2f(st[kb][r]), __builtin_amdgcn_exp2f(st[kb][r + 1]));
                    pk[kb][r >> 1] = w;
                    psum[(r >> 1) & 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, psum[(r >> 1) & 1], false);
                }
            redo = __any(!(psum[0] + psum[1] <= TRIG));   // wave-uniform; also taken for inf / NaN sums
        }
        if (__builtin_expect(redo, FIRST)) {
            // re-reference: true maximum of this tile's (score - m); both halves of a query agree on it
            float mx = -1.0e30f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
            mx = halves_max(mx);
            // new reference (a bf16 value): the first tile takes its maximum whatever the sign, later tiles only move up
            const float want = (FIRST || mx > 0.f) ? m_ref + mx : m_ref;
            const float m_new = bf16_to_f32(f32_to_bf16(fmaxf(want, -3.0e38f)));
            const float delta = m_new - m_ref;              // exact: both are bf16 values
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_ref = m_new;
            qx[0] = (short)(hh == 0 ? f32_to_bf16(-m_new) : 0);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            psum[0] = psum[1] = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const uint32_t w = pack_bf16x2(__builtin_amdgcn_exp2f(st[kb][r] - delta), __builtin_amdgcn_exp2f(st[kb][r + 1] - delta));
                    pk[kb][r >> 1] = w;
                    psum[(r >> 1) & 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, psum[(r >> 1) & 1], false);
                }
        }
        l_run += psum[0] + psum[1];
        // O^T += V^T P^T ; k-slots of step (kb,u): regs 8u..8u+7 <-> keys 32kb+16u+4hh+{0..3, 8..11}
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 pw = {pk[kb][4 * u], pk[kb][4 * u + 1], pk[kb][4 * u + 2], pk[kb][4 * u + 3]};
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int blk = (kb * 32 + 16 * u) * (D * 2);
                    const s16x4_t v_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk));
                    const s16x4_t v_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t *)(uintptr_t)(vs_ + vofs[d] + blk + 8 * (D * 2)));
                    const bf16x8_t vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
    };

    constexpr std::integral_constant<bool, true> CTRUE{};
    constexpr std::integral_constant<bool, false> CFALSE{};
    constexpr std::integral_constant<int, 2> FULL{};
    constexpr std::integral_constant<int, 1> HALF{};
    constexpr std::integral_constant<int, 0> C0{};
    constexpr std::integral_constant<int, 1> C1{};
    constexpr std::integral_constant<int, 2> C2{};
    constexpr std::integral_constant<int, -1> CDYN{};
    const bool live_wave = a.no_trim || a.row0 + qt * QB + wave * 32 < a.S;
    const bool short_tail = !a.no_trim && a.S - (nkt - 1) * KVBLK <= 32;
    stage_kv<D, false, true, NWB>(kb_, a.k_ts, 0, a.S, smem, wave, lane, kvo);
    stage_kv<D, true, true, NWB>(vb_, a.v_ts, 0, a.S, smem + TILE, wave, lane, vvo);
    // (tile_step as in the plain schedule: wait for tile t, start the next one's DMA, then the tile body)
    auto tile_step = [&](int t, auto nkb_, auto stage_, auto next_, auto first_) __attribute__((always_inline)) {
        constexpr int STAGE = decltype(stage_)::value, NEXT = decltype(next_)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int slot = STAGE >= 0 ? STAGE : (t & 1);
        const uint32_t ks_ = (uint32_t)(slot * 2 * TILE);
        if constexpr (NEXT != 0) {
            char *nx = smem + (slot ^ 1) * 2 * TILE;
            stage_kv<D, false, NEXT == 2, NWB>(kb_, a.k_ts, (t + 1) * KVBLK, a.S, nx, wave, lane, kvo);
            stage_kv<D, true, NEXT == 2, NWB>(vb_, a.v_ts, (t + 1) * KVBLK, a.S, nx + TILE, wave, lane, vvo);
        }
        if (live_wave) tile_body(ks_, ks_ + TILE, t * KVBLK, nkb_, std::integral_constant<bool, NEXT != 1>{}, first_);
    };
    // first tile (sets m), steady state unrolled by ring parity, the step that stages the (possibly ragged) last tile, the
    // last tile itself
    if (nkt == 1) {
        if (short_tail) tile_step(0, HALF, C0, C0, CTRUE); else tile_step(0, FULL, C0, C0, CTRUE);
    } else {
        if (nkt == 2) tile_step(0, FULL, C0, C2, CTRUE); else tile_step(0, FULL, C0, C1, CTRUE);
        const int n_main = nkt - 2;
        for (int t = 1; t < n_main; t += 2) {
            tile_step(t, FULL, C1, C1, CFALSE);
            if (t + 1 < n_main) tile_step(t + 1, FULL, C0, C1, CFALSE);
        }
        if (nkt >= 3) tile_step(nkt - 2, FULL, CDYN, C2, CFALSE);
        if (short_tail) tile_step(nkt - 1, HALF, CDYN, C0, CFALSE); else tile_step(nkt - 1, FULL, CDYN, C0, CFALSE);
    }

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


/* launcher branch (inside attn_fwd_launch):
    if (var & 32) var = 64 | 2;   // automatic: the folded schedule (the max subtraction rides on the MFMA)
    a.no_trim = (var >> 4) & 1;
    if (var & 64) {
        // full 256-row blocks (8 waves) first, then the remaining rows with 4-wave blocks; var bit 3: 4-wave blocks only
        const int n8 = (var & 8) ? 0 : a.S / 256;
        if (n8 > 0) {
            AttnArgs a8 = a;
            a8.row0 = 0; a8.nqt = n8;
            const dim3 g8((unsigned)(groups * 8 * n8));
            if (D == 64) VLLM_LAUNCH((attn_fwd_fold_kernel<64, 8>), g8, dim3(512), lds, st, a8);
            else VLLM_LAUNCH((attn_fwd_fold_kernel<128, 8>), g8, dim3(512), lds, st, a8);
            VLLM_CHECK_LAUNCH("attn_fwd_fold_kernel<8 waves>");
        }
        const int rem = a.S - n8 * 256;
        if (rem > 0) {
            AttnArgs a4 = a;
            a4.row0 = n8 * 256; a4.nqt = (rem + QBLK - 1) / QBLK;
            const dim3 g4((unsigned)(groups * 8 * a4.nqt));
            if (D == 64) VLLM_LAUNCH((attn_fwd_fold_kernel<64, 4>), g4, dim3(256), lds, st, a4);
            else VLLM_LAUNCH((attn_fwd_fold_kernel<128, 4>), g4, dim3(256), lds, st, a4);
            VLLM_CHECK_LAUNCH("attn_fwd_fold_kernel<4 waves>");
        }
        return VLLM_OK;
    }
*/
