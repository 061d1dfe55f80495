// Multi-scale deformable attention forward, LDS-tiled kernel, generation 7 ("msda_tiled" 15 / 16; the round-2 default -- round 3's is
// generation 8, msda_tiled8.hip): generation 6's pyramid items (msda_tiled6.hip) in a software pipeline across items.
//
// What the phase clock of generation 6 showed (profiles/r02_msda6_phases.txt): per item a block spends a quarter of its
// time issuing / waiting for the window DMA (a CU pulls ~25 GB/s through the LDS-DMA path, the windows of an item are
// ~66 KB), a third in the gather (LDS-bandwidth bound), a fifth in two barriers -- one after the other, and only the
// second block of the CU can fill the gaps.  Here ONE block of 12 waves owns the CU and an item is one pass of the block
// (12 x 16 = 192 (query, head) slots for its 170 queries):
//   * ONE window arena of 1200 pixels shared by two items, one growing from the bottom and the next from the top: while
//     item n is gathered, the windows of item n + 1 are DMA'd in beside it (an item's windows take ~520 pixels on average,
//     up to ~700: two fixed arenas of 590 pixels left every third item with a level that had to be gathered from global
//     memory; a level that does not fit beside the previous item is staged one item later instead).  The DMA instructions are issued a few at a time BETWEEN the points of the gather (a wave that issues its share in one
//     burst sits in the issue queue for microseconds), so the memory pipe streams while the LDS pipe gathers;
//   * point arithmetic + bounding boxes of item n + 1 are evaluated before item n is gathered, its sampling locations were
//     requested an item earlier: ONE barrier per item (boxes of n + 1 complete, windows of n landed, arena of n - 1 free);
//   * levels of an item that do not fit the arena next to the others (or not at all) are gathered from global memory.
// Everything else -- pyramid items, a lane owns one level of its query, quad per (query, head), box reduction by LDS
// integer minima, the integer part of a point -- is generation 6's and shares its helpers.
//
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "msda_tiled6_helpers.hpp"

// Timing-only ablation builds (tools/msda7_ablate.sh): -DT7_ABL=<mask> removes one cost at a time; results are wrong by
// construction.  1: no multiply-adds in the gather, 2: no LDS reads in the gather, 4: no window DMA, 8: no output stores,
// 16: no gather at all, 64: no per-item barrier, 128: window DMA addresses computed but no load issued, 256: window DMA
// reads the zero line only (issue + LDS write cost without value traffic), 512: no point arithmetic / box reduction (constant points).
#ifndef T7_ABL
#define T7_ABL 0
#endif
#ifndef T7_OLD_PLACEMENT      // 1: a DMA round BETWEEN two gather points (round 2); 0: inside a point, under its LDS reads
#define T7_OLD_PLACEMENT 0
#endif

namespace vllm {

namespace {

__device__ unsigned long long g_t7_prof[16];
#define T7_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// NW waves (NW * 16 >= 170 slots), WIN pixels per arena, ONE block per CU
template <int NW, int WIN, bool PROF>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void msda_fwd_tiled7_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out, uint16_t *__restrict__ out16, int hinted)
{
    constexpr int D = 32, PT = 4, THREADS = NW * 64, QPP = NW * 8;
    static_assert(NW * 16 >= 170, "an item has up to 170 queries");
    constexpr int R = WIN;   // pixels of the window ring (the windows of two consecutive items live in it)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_box = reinterpret_cast<int *>(smem + (T6_ZPX + R + T6_SLACK) * 128);   // [3][4 levels][4]: min hl, min -hl, min wl, min -wl

    if (!geometry_is_nested(shapes, L, Lq)) {
        // hinted: the host said "pyramid" and launched no other kernel -- a stale hint must fail loudly, not leave `out` unwritten
        if (hinted) __builtin_trap();
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned pacc[16] = {};   // (dead in the production instantiation)
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const int k = lane & 3;                       // the value level this lane owns
    const unsigned MD = (unsigned)(M * D);
    const int H0 = (int)shapes[0], W0 = (int)shapes[1];
    // level maps: nested (msda_sample.hpp), not necessarily exact halves -- sizes and query starts from the shape tensor
    const int H1 = L > 1 ? (int)shapes[2] : 1, W1 = L > 1 ? (int)shapes[3] : 1, H2 = L > 2 ? (int)shapes[4] : 1, W2 = L > 2 ? (int)shapes[5] : 1;
    const int H3 = L > 3 ? (int)shapes[6] : 1, W3 = L > 3 ? (int)shapes[7] : 1;
    const int qs1 = H0 * W0, qs2 = qs1 + H1 * W1, qs3 = qs2 + H2 * W2;
    const int ntx0 = (W0 + 15) >> 4;
    const int n_tiles = ((H0 + 7) >> 3) * ntx0;
    const unsigned n_items = (unsigned)(B * M * n_tiles);
    const int n_slots = L == 1 ? 128 : L == 2 ? 160 : L == 3 ? 168 : 170;

    // ---- per-lane constants (as generation 6) ----
    const int quad = lane >> 2;
    const int hf = (quad >> 1) & 1;
    const int cA = (hf * 4 + k) * 16;
    const int cA0 = cA + (int)lds_addr(smem);      // ... as an LDS byte address
    const int qslot = (quad & 8) | ((0x46751320 >> ((quad & 7) * 4)) & 7);
    const int kk = min(k, L - 1);
    const int Hk = sel4(kk, H0, H1, H2, H3), Wk = sel4(kk, W0, W1, W2, W3);
    const int v0k = (int)lsi[kk];
    const int sub8 = lane & 7;                    // DMA: 16-byte chunk of the (pixel, head) row
    int sinfo, sq0, sH, sW;   // this lane's query slot: packed (level, y, x), first query of its level, the level's size
    {
        const int s = wave_s * 16 + qslot;
        const int rr = min((s >= 128) + (s >= 160) + (s >= 168), 3);
        const int lo = s - (rr == 0 ? 0 : rr == 1 ? 128 : rr == 2 ? 160 : 168);
        sinfo = rr | ((lo >> (4 - rr)) << 2) | ((lo & ((16 >> rr) - 1)) << 6) | ((s >= n_slots ? 1 : 0) << 10);
        sq0 = sel4(rr, 0, qs1, qs2, qs3);
        sH = sel4(rr, H0, H1, H2, H3); sW = sel4(rr, W0, W1, W2, W3);
    }

    for (int i = tid; i < T6_ZPX * 32; i += THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    if (tid < 48) s_box[tid] = T6_BIG;
    __syncthreads();

    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;

    auto pair_of = [&](int b, int m, int ty, int tx, bool &ok) -> unsigned {
        const int sr = sinfo & 3, sy = (sinfo >> 2) & 15, sx = (sinfo >> 6) & 15;
        const int y = ((ty * 8) >> sr) + sy, x = ((tx * 16) >> sr) + sx;
        ok = !(sinfo >> 10) && y < sH && x < sW;
        const int q = ok ? sq0 + y * sW + x : (ty * 8) * W0 + tx * 16;
        return (unsigned)((b * Lq + q) * M + m);
    };
    float4_t lc0, lc1, la;   // this lane's level of the item AFTER the one being prepared: 4 x (x, y), 4 weights
    auto fetch = [&](unsigned pair) {
        const unsigned e = (pair * (unsigned)L + (unsigned)kk) * PT;
        lc0 = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2);
        lc1 = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2 + 4);
        la = *reinterpret_cast<const float4_t *>(attw + (size_t)e);
    };
    auto decode = [&](unsigned item, int &b, int &m, int &ty, int &tx) {
        const unsigned bm = item / (unsigned)n_tiles, t = item - bm * (unsigned)n_tiles;
        const unsigned bb = bm / (unsigned)M, yy = t / (unsigned)ntx0;
        b = __builtin_amdgcn_readfirstlane((int)bb); m = __builtin_amdgcn_readfirstlane((int)(bm - bb * (unsigned)M));
        ty = __builtin_amdgcn_readfirstlane((int)yy); tx = __builtin_amdgcn_readfirstlane((int)(t - yy * (unsigned)ntx0));
    };

    unsigned j = blockIdx.x >> 3;
    if (!(j < ipx && xcd * ipx + j < n_items)) return;   // (block-uniform) nothing to do
    // "next" = the item being prepared (points, boxes, windows); "cur" = the item being gathered
    int nb, nm, nty, ntx;
    bool nv = true;                 // the next item exists
    decode(xcd * ipx + j, nb, nm, nty, ntx);
    bool nqok;
    unsigned npr = pair_of(nb, nm, nty, ntx, nqok);
    fetch(npr);

    bool cv = false;                // the current item exists
    float w1c[4] = {}, w2c[4] = {}, w3c[4] = {}, w4c[4] = {};
    int oc[4] = {};
    int4 bxc = {0, 0, 0, 0};
    int layc = 0, cb = 0, cm = 0;
    unsigned prc = 0;
    bool qokc = false;
    int bsel = 0;                   // box buffer of next: (item index in this block's sequence) % 3
    int side = 0;                   // side of the arena NEXT grows from (0 bottom, 1 top); cur sits on the other one
    int used_cur = 0;               // pixels cur's windows take on its side
    int late_c = -1, late_np_c = 0; // cur's late level (or -1) and its window size
    bool late_c_placed = true;
    unsigned okmc = 0;
    while (true) {
        T7_TICK(0)   // loop control, stores of the previous item
        // ---- S1 (next item): this lane's 4 points of level k ----
        float w1n[4], w2n[4], w3n[4], w4n[4];
        int on[4];
        unsigned okm = 0;
        int r0 = T6_BIG, r1 = T6_BIG, r2 = T6_BIG, r3 = T6_BIG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lx = i < 2 ? lc0[2 * i] : lc1[2 * i - 4], ly = i < 2 ? lc0[2 * i + 1] : lc1[2 * i - 3];
            const SamplePoint<float> sp = sample_point<float>(lx, ly, Hk, Wk);
            const bool ok = sp.ok && nqok && k < L && nv;
            const float lh = sp.h_im - (float)sp.h_low, lw = sp.w_im - (float)sp.w_low;
            const float hh = 1.f - lh, hw_ = 1.f - lw;
            const float a = la[i];
            w1n[i] = ok ? (hh * hw_) * a : 0.f; w2n[i] = ok ? (hh * lw) * a : 0.f;
            w3n[i] = ok ? (lh * hw_) * a : 0.f; w4n[i] = ok ? (lh * lw) * a : 0.f;
            on[i] = ((sp.h_low + 1) << 16) | (sp.w_low + 1);
            okm |= ok ? (1u << i) : 0u;
            r0 = min(r0, ok ? sp.h_low : T6_BIG); r1 = min(r1, ok ? -sp.h_low : T6_BIG);
            r2 = min(r2, ok ? sp.w_low : T6_BIG); r3 = min(r3, ok ? -sp.w_low : T6_BIG);
        }
        T7_TICK(1)   // wait for the prefetched locations + point arithmetic
        // ---- S2: boxes of all levels ----
        r0 = dpp_min<0x128>(dpp_min<0x124>(r0)); r1 = dpp_min<0x128>(dpp_min<0x124>(r1));   // row_ror:4, row_ror:8
        r2 = dpp_min<0x128>(dpp_min<0x124>(r2)); r3 = dpp_min<0x128>(dpp_min<0x124>(r3));
        int *boxp = s_box + bsel * 16;
        if ((lane & 12) == 0) {
            const unsigned a = lds_addr(boxp + k * 4);
            asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_min_i32 %0, %3 offset:8\n\tds_min_i32 %0, %4 offset:12"
                         :: "v"(a), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        }
        // the buffer of the item after next: read for the last time two barriers ago, written again behind this barrier
        if (tid < 16) s_box[(bsel == 2 ? 0 : bsel + 1) * 16 + tid] = T6_BIG;
        T7_TICK(2)   // box reduction
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the current item's windows has landed
        T7_TICK(3)   // wait for the own DMA
        if (!(T7_ABL & 64)) __syncthreads();   // (X) boxes of next complete; windows of cur landed; everyone is done gathering the item before cur
        T7_TICK(4)   // barrier X
        const int4 bxn = *reinterpret_cast<const int4 *>(boxp + k * 4);   // lane l < 4: the box of level l
        // ---- S3: window placement (block-uniform).  The arena is shared by TWO items: one grows from the bottom, the next
        //      from the top (cur and next always sit on opposite sides; everyone has finished with the item before cur).
        //      Per level, in the lane that owns it:  base | hot << 24 | cold << 25 | late << 26.  A level of next that does
        //      not fit beside cur's windows becomes LATE (one per item): it is placed and staged an item later, when the
        //      item before it has left the arena, and gathered behind an extra barrier; what cannot be placed even then
        //      (or has a pitch beyond the zero strip) is gathered from global memory.
        const bool anyk = bxn.x != T6_BIG && k < L;
        const int wwk = (-bxn.w + 1) - bxn.z + 1;
        int np8k = anyk ? ((((-bxn.y + 1) - bxn.x + 1) * wwk + 7) & ~7) : 0;
        if (anyk && wwk > T6_ZPX - 2) np8k = 0x10000;
        const unsigned magick = (1u << 20) / (unsigned)max(wwk, 1) + 1u;   // pix / ww for pix * ww < 2^20 (used by the DMA rounds)
        // (a) cur's late level: the other side of the arena is free now
        if (late_c >= 0) {
            const bool fits = used_cur + late_np_c <= R;
            const int base = side ? used_cur : R - used_cur - late_np_c;   // cur sits on side ^ 1
            if (fits) used_cur += late_np_c; else late_c_placed = false;
            if (k == late_c) {
                layc = fits ? (base | (5 << 24)) : (1 << 25);
                if (fits) {
                    const int y0c = bxc.x, x0c = bxc.z, wwc = (-bxc.w + 1) - bxc.z + 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int hl = (oc[i] >> 16) - 1, wl = (oc[i] & 0xffff) - 1;
                        oc[i] = ((okmc >> i) & 1u) ? (T6_ZPX + base + (hl - y0c) * wwc + (wl - x0c)) * 128 : 0;
                    }
                }
            }
        }
        // (b) next's windows beside cur's
        int layn, late_n = -1, late_np_n = 0, used_next = 0;
        int cum[5] = {0, 0, 0, 0, 0};   // pixels of next's HOT windows up to level l (wave-uniform): the DMA rounds walk their concatenation
        {
            const int limit = R - used_cur;
            int lay[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int np = __builtin_amdgcn_readlane(np8k, l);
                const bool fits = np > 0 && used_next + np <= limit;
                const bool late = !fits && np > 0 && np <= R && late_n < 0;
                const int base = side ? R - used_next - np : used_next;
                lay[l] = fits ? (base | (1 << 24)) : late ? (1 << 26) : (np > 0 ? (1 << 25) : 0);
                used_next += fits ? np : 0;
                cum[l + 1] = used_next;
                late_np_n = late ? np : late_np_n;
                late_n = late ? l : late_n;
            }
            layn = sel4(k, lay[0], lay[1], lay[2], lay[3]);
        }
        // ---- S5: LDS byte offsets of next's points ----
        {
            const int y0k = bxn.x, x0k = bxn.z;
            const int basek = layn & 0xffff;
            const bool hotk = (layn >> 24) & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hl = (on[i] >> 16) - 1, wl = (on[i] & 0xffff) - 1;
                const bool use = (okm >> i) & 1u;
                const int off = use ? (T6_ZPX + basek + (hl - y0k) * wwk + (wl - x0k)) * 128 : 0;
                on[i] = hotk ? off : on[i];
            }
        }
        T7_TICK(5)   // layout + offsets
        // ---- the item after next: locations / weights in flight during the gather ----
        const float *vbn = value + ((size_t)nb * S * M + nm) * D;   // next: (b, pixel 0, head m)
        j += blocks_per_xcd;
        const bool av = nv && j < ipx && xcd * ipx + j < n_items;
        int ab = nb, am = nm, aty = nty, atx = ntx;
        if (av) decode(xcd * ipx + j, ab, am, aty, atx);
        bool aqok;
        const unsigned apr = pair_of(ab, am, aty, atx, aqok);
        fetch(apr);
        T7_TICK(6)   // decode + prefetch issue
        // ---- cur's late level: its DMA goes out first, in one burst; it lands while the other levels are gathered ----
        const float *vbc = value + ((size_t)cb * S * M + cm) * D;
        if (late_c >= 0 && late_c_placed && !(T7_ABL & 4)) {
            const int lay_l = __builtin_amdgcn_readlane(layc, late_c);
            const int y0 = __builtin_amdgcn_readlane(bxc.x, late_c), ny1 = __builtin_amdgcn_readlane(bxc.y, late_c);
            const int x0 = __builtin_amdgcn_readlane(bxc.z, late_c), nx1 = __builtin_amdgcn_readlane(bxc.w, late_c);
            const int ww = (-nx1 + 1) - x0 + 1, npix = ((-ny1 + 1) - y0 + 1) * ww;
            const int Hl = sel4(late_c, H0, H1, H2, H3), Wl = sel4(late_c, W0, W1, W2, W3);
            const unsigned magic = (1u << 20) / (unsigned)ww + 1u;
            const float *srcl = vbc + (size_t)__builtin_amdgcn_readlane(v0k, late_c) * MD;
            char *dst = smem + (T6_ZPX + (lay_l & 0xffff)) * 128;
            for (int i0 = wave_s * 8; i0 < npix; i0 += QPP) {
                const int pix = i0 + (lane >> 3);
                const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
                const int gy = y0 + wy, gx = x0 + wx;
                const bool inside = (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl;
                const float *src = inside ? srcl + (size_t)((unsigned)(gy * Wl + gx) * MD) + sub8 * 4 : g_t6_zero_px + sub8 * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(dst + i0 * 128), 16, 0, 0);
            }
        }

        // ---- window DMA of next, one round (8 pixels per wave) per call, issued under the LDS reads of the gather ----
        // Round 3: the hot windows of the item are ONE concatenated list of 8-pixel groups (windows are padded to 8 pixels),
        // group g belongs to wave g % NW; a round is straight-line code on scalars (group -> level by three compares against
        // the cumulative pixel counts, the level's box / base / magic by v_readlane with that scalar) + ~12 VALU for the
        // lane's pixel.  The source is a raw BUFFER load (descriptor = the level's slab of this (batch, head)): a pixel of
        // the out-of-map ring gets an offset beyond the descriptor's size and the hardware returns zeros -- no zero line, no
        // 64-bit address arithmetic, no select between two bases.  (The round-2 form kept per-level loop state that the
        // compiler turned into VGPRs under exec masks: 83 s_and_saveexec in the kernel; ablation builds put the DMA's
        // address work at 88 us of 518, profiles/r03_msda7_ablation.txt.)
        int dg = wave_s;                              // this wave's next group
        const int lpx = lane >> 3;
        // (readfirstlane: every value below IS wave-uniform, but the compiler's divergence analysis loses that somewhere in the
        //  item loop and would keep the state in VGPRs under exec masks and wrap the load in a waterfall loop)
        auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
        auto dma_round = [&]() {
            const int p0 = uni(dg) * 8, c1 = uni(cum[1]), c2 = uni(cum[2]), c3 = uni(cum[3]), c4 = uni(cum[4]);
            if (p0 >= c4) return;                     // (wave-uniform)
            dg += NW;
            if (T7_ABL & 4) return;
            const int l = (p0 >= c1) + (p0 >= c2) + (p0 >= c3);
            const int pix0 = p0 - (l == 0 ? 0 : l == 1 ? c1 : l == 2 ? c2 : c3);
            const int lay_l = __builtin_amdgcn_readlane(layn, l);
            const int y0 = __builtin_amdgcn_readlane(bxn.x, l), x0 = __builtin_amdgcn_readlane(bxn.z, l);
            const int ww = (-__builtin_amdgcn_readlane(bxn.w, l) + 1) - x0 + 1;
            const unsigned magic = (unsigned)__builtin_amdgcn_readlane((int)magick, l);
            const int Hl = uni(sel4(l, H0, H1, H2, H3)), Wl = uni(sel4(l, W0, W1, W2, W3));
            const uint64_t lvl = (uint64_t)(uintptr_t)vbn + (uint64_t)(unsigned)__builtin_amdgcn_readlane(v0k, l) * (uint64_t)uni((int)MD) * 4u;
            const uint64_t lvl_u = ((uint64_t)(unsigned)uni((int)(lvl >> 32)) << 32) | (unsigned)uni((int)(unsigned)lvl);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)lvl_u, 0,
                                                                                 (int)(((unsigned)(Hl * Wl - 1) * (unsigned)uni((int)MD) + 32u) * 4u), 0x00020000);
            const int pix = pix0 + lpx;
            const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
            const int gy = y0 + wy, gx = x0 + wx;
            const bool inside = (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl && !(T7_ABL & 256);
            const unsigned voff = inside ? ((unsigned)(gy * Wl + gx) * MD + (unsigned)sub8 * 4u) * 4u : 0xfffffff0u;
            if (T7_ABL & 128) { asm volatile("" :: "v"(voff)); return; }
            char *dst = smem + (size_t)(T6_ZPX + (lay_l & 0xffff) + pix0) * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, (int)voff, 0, 0, 0);
        };

        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        if (cv) {
            // ---- S6: gather cur.  Level LQ's owner is quad lane LQ ----
            // The eight 16-byte reads of a point are ONE asm statement (reads + their wait): left to the compiler, a ds_read
            // behind an LDS-DMA in flight gets an s_waitcnt vmcnt(0) in front of it (the DMA "may alias" it), which would
            // serialise the next item's window DMA with this item's gather.  Which arena a read touches is ours to know.
            // Round 3: the reads of a point are issued FIRST and everything that does not need their data -- the DPP
            // broadcasts of the point's four weights, and every second point one round of the next item's window DMA (its
            // address arithmetic: ~18 VALU) -- runs in the shadow of the LDS round trip, in front of the wait.  Before, a DMA
            // round sat BETWEEN two points: its VALU delayed the next point's reads by ~70 cycles per wave and the LDS pipe
            // (the unit the gather is bound by) ran dry -- the ablation builds (profiles/r03_msda7_ablation.txt) put the
            // cost of the address arithmetic alone at 88 us of 518, of issuing the DMA at 13 us.
#define T7_HOT_POINT(I_, LQ, MID)                                                                                \
    {                                                                                                            \
        const int b0 = qbi<LQ>(oc[I_]) + cA0, b1 = b0 ^ 64;                                                       \
        const int b0p = b0 + pitch, b1p = b1 + pitch;                                                            \
        float4_t a1, a2, a3, a4, c1, c2, c3, c4;                                                                 \
        if (T7_ABL & 2) asm volatile("; no reads" : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4) : "v"(b0), "v"(b0p), "v"(b1), "v"(b1p)); else \
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\t"                                 \
                     "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:128\n\t"                                 \
                     "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:128\n\t"                               \
                     "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:128"                                   \
                     : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)    \
                     : "v"(b0), "v"(b0p), "v"(b1), "v"(b1p)                                                      \
                     : "memory");                                                                                \
        const float e1 = qbf<LQ>(w1c[I_]), e2 = qbf<LQ>(w2c[I_]), e3 = qbf<LQ>(w3c[I_]), e4 = qbf<LQ>(w4c[I_]);  \
        MID;                                                                                                     \
        /* the data registers are in / out operands of the wait: no consumer (nor a copy) can be placed above it */ \
        if (!(T7_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)"                                                   \
                     : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4) :: "memory"); \
        if (T7_ABL & 1) { acc[0] += a1[0] + a2[1] + a3[2] + a4[3] + e1; acc[4] += c1[0] + c2[1] + c3[2] + c4[3] + e2 + e3 + e4; } else \
        _Pragma("unroll") for (int c = 0; c < 4; c += 2) {                                                       \
            float2_t t = {acc[c], acc[c + 1]};                                                                   \
            t = t6_fma2(e1, (float2_t){a1[c], a1[c + 1]}, t); t = t6_fma2(e2, (float2_t){a2[c], a2[c + 1]}, t);  \
            t = t6_fma2(e3, (float2_t){a3[c], a3[c + 1]}, t); t = t6_fma2(e4, (float2_t){a4[c], a4[c + 1]}, t);  \
            acc[c] = t.x; acc[c + 1] = t.y;                                                                      \
            float2_t u = {acc[4 + c], acc[5 + c]};                                                               \
            u = t6_fma2(e1, (float2_t){c1[c], c1[c + 1]}, u); u = t6_fma2(e2, (float2_t){c2[c], c2[c + 1]}, u);  \
            u = t6_fma2(e3, (float2_t){c3[c], c3[c + 1]}, u); u = t6_fma2(e4, (float2_t){c4[c], c4[c + 1]}, u);  \
            acc[4 + c] = u.x; acc[5 + c] = u.y;                                                                  \
        }                                                                                                        \
        /* pin the sums here: otherwise the multiply-adds are sunk below the DMA round's branches and the loaded */ \
        /* registers of two points are spilled across them */                                                    \
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),   \
                          "+v"(acc[6]), "+v"(acc[7]));                                                           \
    }
#define T7_LEVEL(LQ)                                                                                             \
    if (((__builtin_amdgcn_readlane(layc, LQ) >> 24) & 5) == want) {                                             \
        const int pitch = ((-__builtin_amdgcn_readlane(bxc.w, LQ) + 1) - __builtin_amdgcn_readlane(bxc.z, LQ) + 1) * 128; \
        if (T7_OLD_PLACEMENT) {                                                                                  \
            T7_HOT_POINT(0, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            T7_HOT_POINT(1, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            dma_round();                                                                                         \
            T7_HOT_POINT(2, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            T7_HOT_POINT(3, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            dma_round();                                                                                         \
        } else {                                                                                                 \
            T7_HOT_POINT(0, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            T7_HOT_POINT(1, LQ, dma_round()) __builtin_amdgcn_sched_barrier(0);                                  \
            T7_HOT_POINT(2, LQ, (void)0) __builtin_amdgcn_sched_barrier(0);                                      \
            T7_HOT_POINT(3, LQ, dma_round()) __builtin_amdgcn_sched_barrier(0);                                  \
        }                                                                                                        \
    }
            for (int pass = 0; pass < ((T7_ABL & 16) ? 0 : 2); ++pass) {
                const int want = pass ? 5 : 1;   // hot levels; then the late one, behind its DMA + a barrier
                if (pass) {
                    if (!(late_c >= 0 && late_c_placed)) break;      // (block-uniform)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                T7_LEVEL(0) T7_LEVEL(1) T7_LEVEL(2) T7_LEVEL(3)
            }
#undef T7_LEVEL
#undef T7_HOT_POINT
            T7_TICK(7)   // gather (+ interleaved DMA issue)
            // cold levels of cur: from global memory, the owner lane's point data by ds_bpermute (run-time level)
            for (int l = 0; l < L; ++l) {
                const int lay_l = __builtin_amdgcn_readlane(layc, l);
                if (!((lay_l >> 25) & 1)) continue;
                const int Hc = sel4(l, H0, H1, H2, H3), Wc = sel4(l, W0, W1, W2, W3);
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
            T7_TICK(8)   // cold levels
        }
        // whatever is left of next's windows (all of them for the first item of the block)
        while (__builtin_amdgcn_readfirstlane(dg) * 8 < __builtin_amdgcn_readfirstlane(cum[4])) dma_round();
        T7_TICK(9)   // remaining DMA issue
        if (cv) {
            if (qokc && !(T7_ABL & 8)) {
                if (out16) {   // the caller (the fused layer) wants the bf16 operand of output_proj
                    uint16_t *op = out16 + (size_t)prc * D;
                    uint2_t o1, o2;
                    o1.x = pack_bf16x2(acc[0], acc[1]); o1.y = pack_bf16x2(acc[2], acc[3]);
                    o2.x = pack_bf16x2(acc[4], acc[5]); o2.y = pack_bf16x2(acc[6], acc[7]);
                    *reinterpret_cast<uint2_t *>(op + cA / 4) = o1;
                    *reinterpret_cast<uint2_t *>(op + (cA ^ 64) / 4) = o2;
                } else {
                    float *op = out + (size_t)prc * D;
                    store4(op + cA / 4, (float4_t){acc[0], acc[1], acc[2], acc[3]});
                    store4(op + (cA ^ 64) / 4, (float4_t){acc[4], acc[5], acc[6], acc[7]});
                }
            }
            if (PROF) pacc[14] += 1;
        }
        if (!nv) break;
        // cur <- next, next <- the item after next
        cv = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) { w1c[i] = w1n[i]; w2c[i] = w2n[i]; w3c[i] = w3n[i]; w4c[i] = w4n[i]; oc[i] = on[i]; }
        bxc = bxn; layc = layn; cb = nb; cm = nm; prc = npr; qokc = nqok;
        nb = ab; nm = am; nty = aty; ntx = atx; npr = apr; nqok = aqok; nv = av;
        okmc = okm; used_cur = used_next; late_c = late_n; late_np_c = late_np_n; late_c_placed = true; side ^= 1;
        bsel = bsel == 2 ? 0 : bsel + 1;
    }
    if (PROF && lane == 0) {   // every wave reports (sums over the 12 waves of every block)
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_t7_prof[i], (unsigned long long)pacc[i]);
    }
}

template <int NW, int WIN, bool PROF>
int t7_go(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B, int S,
          int M, int L, int Lq, float *out, uint16_t *out16, int hinted, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    constexpr size_t lds = (size_t)(T6_ZPX + WIN + T6_SLACK) * 128 + 256;
    static_assert(lds <= 163840, "LDS budget");
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled7_kernel<NW, WIN, PROF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    VLLM_LAUNCH((msda_fwd_tiled7_kernel<NW, WIN, PROF>), dim3((cus / 8) * 8), dim3(NW * 64), lds, st, value, shapes, lsi, loc, attw,
                B, S, M, L, Lq, out, out16, hinted);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled7_kernel");
    return VLLM_OK;
}

}  // namespace

int msda_tiled7_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, int prof, hipStream_t st, uint16_t *out16, int hinted)
{
    if (prof) return t7_go<12, 1200, true>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
    return t7_go<12, 1200, false>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
}

int msda7_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t7_prof), sizeof(h)) != hipSuccess) {
        set_error("msda7_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t7_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

#ifdef T7_ABL_ENTRY
// test entry of the ablation builds (tools/msda7_ablate.py)
extern "C" int t7_abl_run(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B,
                          int S, int M, int L, int Lq, float *out, void *stream)
{
    return msda_tiled7_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, 0, (hipStream_t)stream, nullptr, 1);
}
void set_error(const char *, ...) {}
#endif

}  // namespace vllm
