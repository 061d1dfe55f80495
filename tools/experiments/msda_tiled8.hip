// Multi-scale deformable attention forward, LDS-tiled kernel, generation 8 ("msda_tiled" option 18): generation 7's pyramid
// items, processed by TWO TEAMS of six waves that run half a period apart.
//
// What the ablation builds of generation 7 showed (profiles/r03_msda7_ablation.txt): the gather of an item is bound by the
// LDS pipe (8 x ds_read_b128 per point, a third of them bank-conflicted by construction), everything else of an item --
// point arithmetic, boxes, layout, window DMA addresses -- by VALU issue, and a block runs the two one after the other:
// all twelve waves gather, then all twelve do arithmetic, so each pipe idles while the other one works.
// Here the CU still belongs to ONE block of twelve waves, but an item is owned by a TEAM (six waves: 96 (query, head)
// slots, so the 170 queries of an item take two passes) and the teams are half a period out of phase:
//
//      half period     team 0                                        team 1
//      2n              P1 points + boxes of item a | meeting point   G0, G1 gather of item b (both passes)
//                      | P2 layout, offsets, window DMA of item a
//      2n + 1          G0, G1 gather of item a                       P1 | meeting point | P2 of item b'
//
// ONE block barrier per half period (the swap); inside its preparing half a team meets at an LDS counter of its own (the boxes
// need all six waves), so the other team's gather is never held up by it.
// so in every interval one team loads the LDS pipe and the other one the VALU.  The arena of 1200 pixels is shared: team 0
// places its windows from the bottom, team 1 from the top, each beside what the other team's item occupies at that
// moment (s_used); ONE level that does not fit becomes "late" (placed in the space the other team's item frees at the swap, staged at
// the start of the gathering half), any further one is gathered from global memory.  An item's windows
// are requested in P2 and waited for at the barrier behind it (the other team's gather hides the wait for the CU, not for
// the team).  Everything else -- pyramid items, a lane owns one level of its query, quad per (query, head), box reduction
// by LDS integer minima, DMA rounds as scalar code with hardware zero fill -- is generation 7's and shares its helpers.
//
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "msda_tiled6_helpers.hpp"

// Timing-only ablation builds: -DT8_ABL=<mask>.  1: no multiply-adds in the gather, 2: no LDS reads in the gather,
// 4: no window DMA, 8: no output stores, 16: no gather at all, 32: no point arithmetic (P1 skipped).
#ifndef T8_ABL
#define T8_ABL 0
#endif
#ifndef T8_GPRIO      // s_setprio of a wave while it gathers (0: none; measured on one box each: 2: 453 vs 469 us, 3: 466 vs 477)
#define T8_GPRIO 2
#endif
#ifndef T8_EARLY      // levels of pass 0 a team gathers at the end of its preparing half (0: none; measured with 2: 500 vs 465 us -- the eight sums carried across the swap spill)
#define T8_EARLY 0
#endif

namespace vllm {

namespace {

__device__ unsigned long long g_t8_prof[16];
#define T8_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// EXACT: the level maps are exact halves (H_l << l == H_0): sizes by shifts, nothing per level held or loaded -- the kernel sits at
// its SGPR budget, and the general form (sizes of nested maps from LDS) costs it 5 % on the shapes that do not need it.
template <int WIN, bool PROF, bool EXACT>
__global__ __launch_bounds__(768, 1) void msda_fwd_tiled8_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out, uint16_t *__restrict__ out16, int hinted)
{
    constexpr int D = 32, PT = 4, NW = 12, TW = 6, THREADS = NW * 64, R = WIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_box = reinterpret_cast<int *>(smem + (T6_ZPX + R + T6_SLACK) * 128);   // [2 teams][4 levels][4]: min hl, min -hl, min wl, min -wl
    int *s_used = s_box + 32;                                                      // [2 teams]: pixels of the arena the team's item occupies
    int *s_cnt = s_box + 34;                                                       // [2 meeting points][2 teams]: arrivals (monotonic)

    if (EXACT ? !geometry_is_pyramid(shapes, L, Lq) : !(geometry_is_nested(shapes, L, Lq) && !geometry_is_pyramid(shapes, L, Lq))) {
        if (hinted) __builtin_trap();   // a stale "pyramid" hint must fail loudly, not leave `out` unwritten
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave_s >= TW ? 1 : 0;
    const int wt = wave_s - team * TW;            // wave within the team
    unsigned pacc[16] = {};
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const int k = lane & 3;                       // the value level this lane owns
    const unsigned MD = (unsigned)(M * D);
    const int H0 = (int)shapes[0], W0 = (int)shapes[1];
    // level maps: nested (msda_sample.hpp), not necessarily exact halves -- sizes and query starts come from the shape tensor and
    // live in LDS (s_dim: H[4], W[4], first query[4]); eleven more scalars held across the item loop cost the kernel its SGPR budget
    int *s_dim = s_box + 40;
    if (!EXACT && tid < 4) {
        int q0 = 0;
        for (int l = 0; l < tid; ++l) q0 += (l < L) ? (int)shapes[2 * l] * (int)shapes[2 * l + 1] : 0;
        s_dim[tid] = tid < L ? (int)shapes[2 * tid] : 1;
        s_dim[4 + tid] = tid < L ? (int)shapes[2 * tid + 1] : 1;
        s_dim[8 + tid] = q0;
    }
    const int ntx0 = (W0 + 15) >> 4;
    const int n_tiles = ((H0 + 7) >> 3) * ntx0;
    const unsigned n_items = (unsigned)(B * M * n_tiles);
    const int n_slots = L == 1 ? 128 : L == 2 ? 160 : L == 3 ? 168 : 170;

    // ---- per-lane constants (as generation 6) ----
    const int quad = lane >> 2;
    const int hf = (quad >> 1) & 1;
    const int cA = (hf * 4 + k) * 16;
    const int cA0 = cA + (int)lds_addr(smem);
    const int qslot = (quad & 8) | ((0x46751320 >> ((quad & 7) * 4)) & 7);
    const int kk = min(k, L - 1);
    const int Hk = EXACT ? H0 >> kk : (int)shapes[2 * kk], Wk = EXACT ? W0 >> kk : (int)shapes[2 * kk + 1];
    const int v0k = (int)lsi[kk];
    const int sub8 = lane & 7;
    int sinfo[2];   // per pass, this lane's query slot: packed (level, y, x, dead)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int s = p * (TW * 16) + wt * 16 + qslot;
        const int rr = min((s >= 128) + (s >= 160) + (s >= 168), 3);
        const int lo = s - (rr == 0 ? 0 : rr == 1 ? 128 : rr == 2 ? 160 : 168);
        sinfo[p] = rr | ((lo >> (4 - rr)) << 2) | ((lo & ((16 >> rr) - 1)) << 6) | ((s >= n_slots ? 1 : 0) << 10);
    }

    for (int i = tid; i < T6_ZPX * 32; i += THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    if (tid < 32) s_box[tid] = T6_BIG;
    if (tid < 2) s_used[tid] = 0;
    if (tid < 6) s_cnt[tid] = 0;
    __syncthreads();

    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;
    const unsigned j0 = blockIdx.x >> 3;
    // items of this block: xcd * ipx + j0 + i * blocks_per_xcd, i = 0 .. n_blk - 1; team t owns the i with i % 2 == t
    const unsigned lim = xcd * ipx < n_items ? min(ipx, n_items - xcd * ipx) : 0u;
    const int n_blk = j0 < lim ? (int)((lim - j0 + blocks_per_xcd - 1) / blocks_per_xcd) : 0;
    if (n_blk == 0) return;   // (block-uniform)
    const int nsteps = 2 * ((n_blk + 1) >> 1) + 1;   // half periods: team 0 prepares in the even ones and gathers in the odd ones, team 1 the other way round

    auto pair_of = [&](int p, int b, int m, int ty, int tx, bool &ok) -> unsigned {
        const int sr = sinfo[p] & 3, sy = (sinfo[p] >> 2) & 15, sx = (sinfo[p] >> 6) & 15;
        const int y = ((ty * 8) >> sr) + sy, x = ((tx * 16) >> sr) + sx;
        const int HW0 = H0 * W0;
        const int sH = EXACT ? H0 >> sr : s_dim[sr], sW = EXACT ? W0 >> sr : s_dim[4 + sr];   // (once per item and pass)
        const int sQ = EXACT ? (sr >= 1 ? HW0 : 0) + (sr >= 2 ? HW0 >> 2 : 0) + (sr >= 3 ? HW0 >> 4 : 0) : s_dim[8 + sr];
        ok = !(sinfo[p] >> 10) && y < sH && x < sW;
        const int q = ok ? sQ + y * sW + x : (ty * 8) * W0 + tx * 16;
        return (unsigned)((b * Lq + q) * M + m);
    };
    auto decode = [&](unsigned item, int &b, int &m, int &ty, int &tx) {
        const unsigned bm = item / (unsigned)n_tiles, t = item - bm * (unsigned)n_tiles;
        const unsigned bb = bm / (unsigned)M, yy = t / (unsigned)ntx0;
        b = __builtin_amdgcn_readfirstlane((int)bb); m = __builtin_amdgcn_readfirstlane((int)(bm - bb * (unsigned)M));
        ty = __builtin_amdgcn_readfirstlane((int)yy); tx = __builtin_amdgcn_readfirstlane((int)(t - yy * (unsigned)ntx0));
    };
    auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };

    // ---- "next": the team's item whose locations / weights are in flight ----
    int i_next = team;
    bool nv = false;
    int nb = 0, nm = 0;
    unsigned npr[2] = {0, 0};
    bool nqok[2] = {false, false};
    float4_t lc0[2], lc1[2], la[2];
    auto prefetch_next = [&]() {
        nv = uni(i_next) < n_blk;
        if (nv) {
            int ty, tx;
            decode(xcd * ipx + j0 + (unsigned)uni(i_next) * blocks_per_xcd, nb, nm, ty, tx);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                npr[p] = pair_of(p, nb, nm, ty, tx, nqok[p]);
                const unsigned e = (npr[p] * (unsigned)L + (unsigned)kk) * PT;
                lc0[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2);
                lc1[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2 + 4);
                la[p] = *reinterpret_cast<const float4_t *>(attw + (size_t)e);
            }
        }
        i_next += 2;
    };

    // ---- "cur": the team's item between P1 and G1 ----
    bool cv = false;
    int cb = 0, cm = 0;
    unsigned prc[2] = {0, 0};
    bool qokc[2] = {false, false};
    float w1[2][4] = {}, w2[2][4] = {}, w3[2][4] = {}, w4[2][4] = {};
    int o[2][4] = {};
    unsigned okm[2] = {0, 0};
    int4 bx = {0, 0, 0, 0};
    int lay = 0;

    // gather of one pass of the 16 (query, head) slots of this wave.  want = 1: the levels staged with the item (+ the levels that
    // come from global memory), want = 5: the item's late level (staged at the start of the gathering half, see P2).
    auto gather = [&](const float (&w1c)[4], const float (&w2c)[4], const float (&w3c)[4], const float (&w4c)[4], const int (&oc)[4],
                      float (&acc)[8], int want, int lmin, int lmax) {
        const float *vbc = value + ((size_t)cb * S * M + cm) * D;
#define T8_HOT_POINT(I_, LQ)                                                                                     \
    {                                                                                                            \
        const int b0 = qbi<LQ>(oc[I_]) + cA0, b1 = b0 ^ 64;                                                       \
        const int b0p = b0 + pitch, b1p = b1 + pitch;                                                            \
        float4_t a1, a2, a3, a4, c1, c2, c3, c4;                                                                 \
        if (T8_ABL & 2) asm volatile("; no reads" : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4) : "v"(b0), "v"(b0p), "v"(b1), "v"(b1p)); else \
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\t"                                 \
                     "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:128\n\t"                                 \
                     "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:128\n\t"                               \
                     "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:128"                                   \
                     : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)    \
                     : "v"(b0), "v"(b0p), "v"(b1), "v"(b1p)                                                      \
                     : "memory");                                                                                \
        const float e1 = qbf<LQ>(w1c[I_]), e2 = qbf<LQ>(w2c[I_]), e3 = qbf<LQ>(w3c[I_]), e4 = qbf<LQ>(w4c[I_]);  \
        if (!(T8_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)"                                                   \
                     : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4) :: "memory"); \
        if (T8_ABL & 1) { acc[0] += a1[0] + a2[1] + a3[2] + a4[3] + e1; acc[4] += c1[0] + c2[1] + c3[2] + c4[3] + e2 + e3 + e4; } else \
        {   /* four independent chains, interleaved corner by corner (round 4: a chain at a time every packed multiply-add   */ \
            /* waits for the one in front of it); per channel the corners are added in the same order as before: same bits */ \
            float2_t t0 = {acc[0], acc[1]}, t1 = {acc[2], acc[3]}, u0 = {acc[4], acc[5]}, u1 = {acc[6], acc[7]};             \
            t0 = t6_fma2(e1, (float2_t){a1[0], a1[1]}, t0); u0 = t6_fma2(e1, (float2_t){c1[0], c1[1]}, u0);                  \
            t1 = t6_fma2(e1, (float2_t){a1[2], a1[3]}, t1); u1 = t6_fma2(e1, (float2_t){c1[2], c1[3]}, u1);                  \
            t0 = t6_fma2(e2, (float2_t){a2[0], a2[1]}, t0); u0 = t6_fma2(e2, (float2_t){c2[0], c2[1]}, u0);                  \
            t1 = t6_fma2(e2, (float2_t){a2[2], a2[3]}, t1); u1 = t6_fma2(e2, (float2_t){c2[2], c2[3]}, u1);                  \
            t0 = t6_fma2(e3, (float2_t){a3[0], a3[1]}, t0); u0 = t6_fma2(e3, (float2_t){c3[0], c3[1]}, u0);                  \
            t1 = t6_fma2(e3, (float2_t){a3[2], a3[3]}, t1); u1 = t6_fma2(e3, (float2_t){c3[2], c3[3]}, u1);                  \
            t0 = t6_fma2(e4, (float2_t){a4[0], a4[1]}, t0); u0 = t6_fma2(e4, (float2_t){c4[0], c4[1]}, u0);                  \
            t1 = t6_fma2(e4, (float2_t){a4[2], a4[3]}, t1); u1 = t6_fma2(e4, (float2_t){c4[2], c4[3]}, u1);                  \
            acc[0] = t0.x; acc[1] = t0.y; acc[2] = t1.x; acc[3] = t1.y; acc[4] = u0.x; acc[5] = u0.y; acc[6] = u1.x; acc[7] = u1.y; \
        }                                                                                                        \
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),   \
                          "+v"(acc[6]), "+v"(acc[7]));                                                           \
    }
#define T8_LEVEL(LQ)                                                                                             \
    if (LQ >= lmin && LQ < lmax && ((__builtin_amdgcn_readlane(lay, LQ) >> 24) & 5) == want) {                   \
        const int pitch = ((-__builtin_amdgcn_readlane(bx.w, LQ) + 1) - __builtin_amdgcn_readlane(bx.z, LQ) + 1) * 128; \
        T8_HOT_POINT(0, LQ) __builtin_amdgcn_sched_barrier(0);                                                   \
        T8_HOT_POINT(1, LQ) __builtin_amdgcn_sched_barrier(0);                                                   \
        T8_HOT_POINT(2, LQ) __builtin_amdgcn_sched_barrier(0);                                                   \
        T8_HOT_POINT(3, LQ) __builtin_amdgcn_sched_barrier(0);                                                   \
    }
        if (!(T8_ABL & 16)) { T8_LEVEL(0) T8_LEVEL(1) T8_LEVEL(2) T8_LEVEL(3) }
#undef T8_LEVEL
#undef T8_HOT_POINT
        // cold levels: from global memory, the owner lane's point data by ds_bpermute (run-time level)
        for (int l = 0; l < (want == 1 && lmax == 4 ? L : 0); ++l) {
            const int lay_l = __builtin_amdgcn_readlane(lay, l);
            if (!((lay_l >> 25) & 1)) continue;
            const int Hc = EXACT ? H0 >> l : uni(s_dim[l]), Wc = EXACT ? W0 >> l : uni(s_dim[4 + l]);
            const float *vc = vbc + (size_t)__builtin_amdgcn_readlane(v0k, l) * MD;
            const int src = ((lane & ~3) | l) << 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hwp = __builtin_amdgcn_ds_bpermute(src, oc[i]);
                const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w1c[i])));
                const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w2c[i])));
                const float e3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w3c[i])));
                const float e4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w4c[i])));
                const int bh = (hwp >> 16) - 1, bw = (hwp & 0xffff) - 1;
                const bool u0 = bh >= 0, u1 = bh + 1 <= Hc - 1, l0 = bw >= 0, l1 = bw + 1 <= Wc - 1;
                const int h0 = min(max(bh, 0), Hc - 1), h1 = min(max(bh + 1, 0), Hc - 1);
                const int c0 = min(max(bw, 0), Wc - 1), c1 = min(max(bw + 1, 0), Wc - 1);
                const float *p1 = vc + (size_t)((unsigned)(h0 * Wc + c0) * MD), *p2 = vc + (size_t)((unsigned)(h0 * Wc + c1) * MD);
                const float *p3 = vc + (size_t)((unsigned)(h1 * Wc + c0) * MD), *p4 = vc + (size_t)((unsigned)(h1 * Wc + c1) * MD);
                const int eA = cA / 4, eB = (cA ^ 64) / 4;
                const float4_t a1 = load4(p1 + eA), a2 = load4(p2 + eA), a3 = load4(p3 + eA), a4 = load4(p4 + eA);
                const float4_t d1 = load4(p1 + eB), d2 = load4(p2 + eB), d3 = load4(p3 + eB), d4 = load4(p4 + eB);
                const float f1 = (u0 && l0) ? e1 : 0.f, f2 = (u0 && l1) ? e2 : 0.f, f3 = (u1 && l0) ? e3 : 0.f, f4 = (u1 && l1) ? e4 : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] += f1 * ((u0 && l0) ? a1[c] : 0.f) + f2 * ((u0 && l1) ? a2[c] : 0.f) +
                              f3 * ((u1 && l0) ? a3[c] : 0.f) + f4 * ((u1 && l1) ? a4[c] : 0.f);
                    acc[4 + c] += f1 * ((u0 && l0) ? d1[c] : 0.f) + f2 * ((u0 && l1) ? d2[c] : 0.f) +
                                  f3 * ((u1 && l0) ? d3[c] : 0.f) + f4 * ((u1 && l1) ? d4[c] : 0.f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PROF) pacc[12] += 1;
        }
    };
    auto store_out = [&](const float (&acc)[8], bool qok, unsigned pr) {
        if (qok && !(T8_ABL & 8)) {
            if (out16) {   // the caller (the fused layer) wants the bf16 operand of output_proj
                uint16_t *op = out16 + (size_t)pr * D;
                uint2_t o1, o2;
                o1.x = pack_bf16x2(acc[0], acc[1]); o1.y = pack_bf16x2(acc[2], acc[3]);
                o2.x = pack_bf16x2(acc[4], acc[5]); o2.y = pack_bf16x2(acc[6], acc[7]);
                *reinterpret_cast<uint2_t *>(op + cA / 4) = o1;
                *reinterpret_cast<uint2_t *>(op + (cA ^ 64) / 4) = o2;
            } else {
                float *op = out + (size_t)pr * D;
                store4(op + cA / 4, (float4_t){acc[0], acc[1], acc[2], acc[3]});
                store4(op + (cA ^ 64) / 4, (float4_t){acc[4], acc[5], acc[6], acc[7]});
            }
        }
    };
    // window DMA of ONE level: its np pixels (a multiple of 8) to arena pixel `base`, in 8-pixel groups (1 KiB per wave instruction);
    // this wave takes the groups g with (g + phase) % 6 == its index in the team.  Everything per level -- box, pitch, magic,
    // buffer descriptor (out-of-map pixels get an offset beyond it: hardware zero fill) -- is set up once, the round itself is
    // ~12 VALU on the lane's pixel + the load.
    auto dma_level = [&](int l, int np, int base, const float *vb, unsigned magick_, int phase) {
        const int y0 = __builtin_amdgcn_readlane(bx.x, l), x0 = __builtin_amdgcn_readlane(bx.z, l);
        const int ww = (-__builtin_amdgcn_readlane(bx.w, l) + 1) - x0 + 1;
        const unsigned magic = (unsigned)__builtin_amdgcn_readlane((int)magick_, l);
        const int Hl = EXACT ? uni(H0) >> l : uni(s_dim[l]), Wl = EXACT ? uni(W0) >> l : uni(s_dim[4 + l]);
        const uint64_t lvl = (uint64_t)(uintptr_t)vb + (uint64_t)(unsigned)__builtin_amdgcn_readlane(v0k, l) * (uint64_t)uni((int)MD) * 4u;
        const uint64_t lvl_u = ((uint64_t)(unsigned)uni((int)(lvl >> 32)) << 32) | (unsigned)uni((int)(unsigned)lvl);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)lvl_u, 0,
                                                                             (int)(((unsigned)(Hl * Wl - 1) * (unsigned)uni((int)MD) + 32u) * 4u), 0x00020000);
        int g0 = wt - phase;            // first group of this wave: (g0 + phase) % 6 == wt
        g0 += g0 < 0 ? TW : 0;
        char *dst0 = smem + (size_t)(T6_ZPX + base) * 128;
        for (int pix0 = uni(g0) * 8; pix0 < np; pix0 += TW * 8) {
            const int pix = pix0 + (lane >> 3);
            const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
            const int gy = y0 + wy, gx = x0 + wx;
            const bool inside = (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl;
            const unsigned voff = inside ? ((unsigned)(gy * Wl + gx) * MD + (unsigned)sub8 * 4u) * 4u : 0xfffffff0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst0 + pix0 * 128), 16, (int)voff, 0, 0, 0);
        }
    };

    // a team's meeting point: wait until `target` arrivals have been counted.  Bounded: the six waves of a team always take the same
    // path, so the count always comes -- but a mistake here must end as a failed launch, not as a hung device
    auto meet = [&](int *cnt, int target) {
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) __builtin_trap();
        }
    };
    int ph = team;                // 0: the preparing half (P1, P2), 1: the gathering half (G0, G1); team 1 enters gathering, with nothing in hand
    int epoch = 0;                // items this team has prepared
    int epoch_late = 0;           // ... of which had a late level
    float acc0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // pass 0's sums of the levels gathered before the swap
    int late_l = -1, late_np = 0, late_base = 0;   // the current item's late level (or -1), its window size and arena position
    unsigned magick_c = 0;        // pix / ww magic of this lane's level (window DMA), kept for the late level
    if (team == 0) prefetch_next();
    for (int step = 0; step < nsteps; ++step) {
        T8_TICK(0)   // barrier + loop control
        const int phu = uni(ph);
        if (phu == 0) {
            // ================= P1: cur <- next; this lane's 4 points of level k, both passes; boxes =================
            cv = nv; cb = nb; cm = nm;
            prc[0] = npr[0]; prc[1] = npr[1]; qokc[0] = nqok[0]; qokc[1] = nqok[1];
            if (cv && !(T8_ABL & 32)) {
                int r0 = T6_BIG, r1 = T6_BIG, r2 = T6_BIG, r3 = T6_BIG;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    unsigned okm_ = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lx = i < 2 ? lc0[p][2 * i] : lc1[p][2 * i - 4], ly = i < 2 ? lc0[p][2 * i + 1] : lc1[p][2 * i - 3];
                        const SamplePoint<float> sp = sample_point<float>(lx, ly, Hk, Wk);
                        const bool ok = sp.ok && qokc[p] && k < L;
                        const float lh = sp.h_im - (float)sp.h_low, lw = sp.w_im - (float)sp.w_low;
                        const float hh = 1.f - lh, hw_ = 1.f - lw;
                        const float a = la[p][i];
                        w1[p][i] = ok ? (hh * hw_) * a : 0.f; w2[p][i] = ok ? (hh * lw) * a : 0.f;
                        w3[p][i] = ok ? (lh * hw_) * a : 0.f; w4[p][i] = ok ? (lh * lw) * a : 0.f;
                        o[p][i] = ((sp.h_low + 1) << 16) | (sp.w_low + 1);
                        okm_ |= ok ? (1u << i) : 0u;
                        r0 = min(r0, ok ? sp.h_low : T6_BIG); r1 = min(r1, ok ? -sp.h_low : T6_BIG);
                        r2 = min(r2, ok ? sp.w_low : T6_BIG); r3 = min(r3, ok ? -sp.w_low : T6_BIG);
                    }
                    okm[p] = okm_;
                }
                r0 = dpp_min<0x128>(dpp_min<0x124>(r0)); r1 = dpp_min<0x128>(dpp_min<0x124>(r1));   // row_ror:4, row_ror:8
                r2 = dpp_min<0x128>(dpp_min<0x124>(r2)); r3 = dpp_min<0x128>(dpp_min<0x124>(r3));
                if ((lane & 12) == 0) {
                    const unsigned a = lds_addr(s_box + team * 16 + k * 4);
                    asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_min_i32 %0, %3 offset:8\n\tds_min_i32 %0, %4 offset:12"
                                 :: "v"(a), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
                }
            }
            T8_TICK(1)
            // ================= P2: layout beside the other team's item, LDS offsets, window DMA =================
            if (cv) {
                // the team's own meeting point (the boxes of all six waves are in): an LDS counter, not the block barrier -- the
                // other team is in the middle of its gather and must not be held up.  The LDS executes a wave's operations in
                // order, so a wave that sees the full count also sees every minimum that was issued in front of an arrival.
                ++epoch;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(s_cnt + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                meet(s_cnt + team, epoch * TW);
                asm volatile("" ::: "memory");
                T8_TICK(8)   // team meeting point
                bx = *reinterpret_cast<const int4 *>(s_box + team * 16 + k * 4);   // lane l < 4: the box of level l
                const int used_other = uni(s_used[team ^ 1]);
                const bool anyk = bx.x != T6_BIG && k < L;
                const int wwk = (-bx.w + 1) - bx.z + 1;
                int np8k = anyk ? ((((-bx.y + 1) - bx.x + 1) * wwk + 7) & ~7) : 0;
                if (anyk && wwk > T6_ZPX - 2) np8k = 0x10000;
                const unsigned magick = (1u << 20) / (unsigned)max(wwk, 1) + 1u;
                int cum[5] = {0, 0, 0, 0, 0};
                magick_c = magick;
                {
                    // Levels are placed on the team's side of the arena beside what the other team's item occupies NOW.  ONE level
                    // that does not fit there becomes LATE: it is placed behind the team's other levels in the space the other
                    // team's item leaves at the swap, reserved here (s_used) so that the other team's next layout keeps clear of
                    // it, staged at the start of the gathering half and gathered behind a second meeting point.
                    const int limit = R - used_other;
                    int used = 0, lays[4];
                    late_l = -1; late_np = 0;
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int np = __builtin_amdgcn_readlane(np8k, l);
                        const bool fits = np > 0 && used + np <= limit;
                        const bool late = !fits && np > 0 && np <= R && late_l < 0;
                        const int base = team ? R - used - np : used;
                        lays[l] = fits ? (base | (1 << 24)) : (np > 0 ? (1 << 25) : 0);
                        used += fits ? np : 0;
                        cum[l + 1] = used;
                        late_np = late ? np : late_np;
                        late_l = late ? l : late_l;
                    }
                    if (late_l >= 0 && used + late_np <= R) {
                        late_base = team ? R - used - late_np : used;
                        const int v = late_base | (5 << 24);
                        if (late_l == 0) lays[0] = v; else if (late_l == 1) lays[1] = v; else if (late_l == 2) lays[2] = v; else lays[3] = v;
                        used += late_np;
                    } else {
                        late_l = -1;
                    }
                    lay = sel4(k, lays[0], lays[1], lays[2], lays[3]);
                    if (tid == team * (TW * 64)) s_used[team] = used;
                }
                {
                    const int y0k = bx.x, x0k = bx.z;
                    const int basek = lay & 0xffff;
                    const bool hotk = (lay >> 24) & 1;
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int hl = (o[p][i] >> 16) - 1, wl = (o[p][i] & 0xffff) - 1;
                            const bool use = (okm[p] >> i) & 1u;
                            const int off = use ? (T6_ZPX + basek + (hl - y0k) * wwk + (wl - x0k)) * 128 : 0;
                            o[p][i] = hotk ? off : o[p][i];
                        }
                }
                T8_TICK(2)   // layout + offsets
                // window DMA: the hot windows are ONE concatenated list of 8-pixel groups, group g belongs to wave g % 6 of the team
                if (!(T8_ABL & 4)) {
                    const float *vbn = value + ((size_t)cb * S * M + cm) * D;
                    // the hot windows, level by level; the groups of the concatenated list are dealt to the six waves round robin
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int np = uni(cum[l + 1]) - uni(cum[l]);
                        if (np > 0) dma_level(l, np, __builtin_amdgcn_readlane(lay, l) & 0xffff, vbn, magick, (uni(cum[l]) >> 3) % TW);
                    }
                }
                T8_TICK(3)   // DMA issue
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the windows has landed
                T8_TICK(4)   // DMA wait
                // The preparing half is the shorter one (the other team's two gather passes set the pace): the team starts on its
                // own gather as soon as ITS windows are in -- a second meeting point of the team, no block barrier: the first
                // T8_EARLY levels of pass 0 (the staged ones; a late level and anything from global memory wait for the swap).
                if (T8_EARLY > 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(s_cnt + 4 + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    meet(s_cnt + 4 + team, epoch * TW);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc0[c] = 0.f;
                    gather(w1[0], w2[0], w3[0], w4[0], o[0], acc0, 1, 0, T8_EARLY);
                    T8_TICK(10)   // early gather
                }
            }
        } else {
            // ================= G0, G1: the two passes of the gather =================
            if (tid - team * (TW * 64) < 16) s_box[team * 16 + (tid - team * (TW * 64))] = T6_BIG;   // read in P2, written again in the next P1
            T8_TICK(5)
            if (cv) {
                if (T8_GPRIO) __builtin_amdgcn_s_setprio(T8_GPRIO);   // the gathering half is the longer one: its waves go first on a SIMD they share
                const bool has_late = uni(late_l) >= 0;   // (team-uniform)
                if (has_late && !(T8_ABL & 4)) {          // its DMA goes out first and lands under pass 0 of the other levels
                    const float *vbc = value + ((size_t)cb * S * M + cm) * D;
                    dma_level(uni(late_l), uni(late_np), uni(late_base), vbc, magick_c, 0);
                }
                float acc[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = T8_EARLY > 0 ? acc0[c] : 0.f;
                gather(w1[0], w2[0], w3[0], w4[0], o[0], acc, 1, T8_EARLY, 4);
                T8_TICK(6)
                if (has_late) {
                    ++epoch_late;
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the late window has landed
                    if (lane == 0) __hip_atomic_fetch_add(s_cnt + 2 + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    meet(s_cnt + 2 + team, epoch_late * TW);
                    asm volatile("" ::: "memory");
                    gather(w1[0], w2[0], w3[0], w4[0], o[0], acc, 5, 0, 4);
                    if (PROF) pacc[13] += 1;
                }
                store_out(acc, qokc[0], prc[0]);
                T8_TICK(9)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = 0.f;
                gather(w1[1], w2[1], w3[1], w4[1], o[1], acc, 1, 0, 4);
                if (has_late) gather(w1[1], w2[1], w3[1], w4[1], o[1], acc, 5, 0, 4);
                store_out(acc, qokc[1], prc[1]);
                if (T8_GPRIO) __builtin_amdgcn_s_setprio(0);
            }
            prefetch_next();   // the team's next item: its locations / weights travel across the barrier into P1
            if (PROF && cv) pacc[14] += 1;
            T8_TICK(7)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        ph ^= 1;
    }
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_t8_prof[i], (unsigned long long)pacc[i]);
    }
}

template <int WIN, bool PROF, bool EXACT>
int t8_go(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B, int S,
          int M, int L, int Lq, float *out, uint16_t *out16, int hinted, hipStream_t st)
{
    const int cus = device_cus();
    constexpr size_t lds = (size_t)(T6_ZPX + WIN + T6_SLACK) * 128 + 256;
    static_assert(lds <= 163840, "LDS budget");
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled8_kernel<WIN, PROF, EXACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    VLLM_LAUNCH((msda_fwd_tiled8_kernel<WIN, PROF, EXACT>), dim3((cus / 8) * 8), dim3(768), lds, st, value, shapes, lsi, loc, attw,
                B, S, M, L, Lq, out, out16, hinted);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled8_kernel");
    return VLLM_OK;
}

}  // namespace

// which: 1 the exact-pyramid instantiation, 2 the nested-maps one (each returns at once on maps that are not its own), 3 both
int msda_tiled8_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, int prof, hipStream_t st, uint16_t *out16, int hinted, int which)
{
    if (which & 1) {
        const int e = prof ? t8_go<1200, true, true>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st)
                           : t8_go<1200, false, true>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
        if (e) return e;
    }
    if (which & 2) {
        const int e = prof ? t8_go<1200, true, false>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st)
                           : t8_go<1200, false, false>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
        if (e) return e;
    }
    return VLLM_OK;
}

int msda8_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t8_prof), sizeof(h)) != hipSuccess) {
        set_error("msda8_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t8_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

#ifdef T8_ABL_ENTRY
extern "C" int t8_abl_run(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B,
                          int S, int M, int L, int Lq, float *out, void *stream)
{
    return msda_tiled8_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, 0, (hipStream_t)stream, nullptr, 1, 1);
}
void set_error(const char *, ...) {}
#endif

}  // namespace vllm
