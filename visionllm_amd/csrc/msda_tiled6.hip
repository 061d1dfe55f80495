// Multi-scale deformable attention forward, LDS-tiled kernel, generation 6 ("msda_tiled" options 10-14 / 17; it serves the bf16-value
// operator vllm_msda_forward_bf16; for fp32 values the automatic choice is generation 8 since round 3, msda_tiled8.hip).
//
// Generation 4 (msda_tiled4.hip) spends two thirds of its time in the per-(tile, level) skeleton: point arithmetic,
// bounding-box reduction + exchange, two barriers, one DMA wait -- four times per item.  This kernel does that work ONCE
// per item and shares the staged windows between more queries:
//   * PYRAMID ITEMS.  When the level maps form an exact 2x pyramid (H_l * 2^l == H_0, FPN strides 8/16/32/64) an item is an
//     8 x 16 pixel region of level 0 of one (batch, head) together with the queries of EVERY level whose cell centre lies
//     in it (128 + 32 + 8 + 2 = 170 queries).  They sample the same neighbourhood of every value level, so one window per
//     value level serves all of them: the 28 % of generation-4 items that tiled the coarse query levels -- the ones with
//     the largest fine-level windows, mostly gathered from global memory by the cold path -- disappear, and the staged
//     volume drops with them.  Other geometries fall back to one 8 x 16 query tile of one level per item ("flat").
//   * ALL LEVELS AT ONCE.  A lane owns ONE value level (lane & 3) of its queries: it loads that level's four sampling
//     locations / weights with two 16-byte loads + one (a (query, head) pair's 128 + 64 bytes are read by the 4 lanes of
//     a quad, fully coalesced), evaluates them, and the bounding boxes of all levels are reduced together: two DPP steps +
//     four LDS integer atomics, ONE barrier.  The windows of all levels are then laid out back to back in one LDS arena and
//     staged by one LDS-DMA phase (levels that do not fit together are split into groups; a level that does not fit at all,
//     or has no accepted point, is gathered from global memory / skipped).  Per item: two barriers instead of eight.
//   * QUAD PER (query, head).  Four lanes x two 16-byte chunks cover the 32 channels; the level's owner lane K broadcasts
//     offset and weights by DPP quad_perm fused INTO the consuming instruction (v_add_u32_dpp for the address, and
//     v_fmac_f32_dpp for every multiply-add: no separate broadcast moves, no packed-math pairing).
//   * LDS BANKS.  A ds_read_b128 is served in 16-lane groups = 4 quads = 4 segments of 64 bytes.  Quads alternate which
//     half of the 128-byte pixel row they read first, and the query <-> quad map pairs the two quads that read the same
//     half with queries whose x differs in bits 0, 1 and 2, so for smooth offset fields their pixels fall into different
//     bank halves at every level.
// The integer part of a sampling point is msda_sample.hpp's sample_point() as everywhere else (index-exact contract); the
// weighted sum is associated as in generation 4 (sum_c (w_c * a) * v_c, levels and points in reference order).
//
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
// VT = float (operator ABI, B3) or bf16 (value written by the fused layer's value_proj epilogue; converted to fp32 while it
// is staged, so the gather is the same); OT likewise.
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "msda_tiled6_helpers.hpp"

namespace vllm {

namespace {

struct T6Item { int b, m, lq, ty, tx; };   // batch, head, (flat: query level), tile row / column

// ---- window staging -------------------------------------------------------------------------------------------------
// fp32 value: LDS-DMA, 8 lanes x 16 B = one 128-byte (pixel, head) row per lane group, NW * 8 pixels per round of the block.
// Ring pixels (row -1 / H, column -1 / W) come from a 128-byte zero line in global memory.
template <int NW>
__device__ __forceinline__ void stage_window(const float *vl, char *dst, int y0, int x0, int wh, int ww, int H, int W,
                                             unsigned MD, int wave_s, int lane)
{
    constexpr int QPP = NW * 8;
    const int npix = wh * ww;
    const int sub = lane & 7;
    const unsigned magic = (1u << 20) / (unsigned)ww + 1u;      // pix / ww for pix * ww < 2^20
    const int dq = (int)(((unsigned)QPP * magic) >> 20), dr = QPP - dq * ww;
    const int pix = wave_s * 8 + (lane >> 3);
    const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
    int gy = y0 + wy, gx = x0 + wx;
    const int xend = x0 + ww;
    const char *vlb = reinterpret_cast<const char *>(vl + sub * 4);
    float *win = reinterpret_cast<float *>(dst);
    if (y0 >= 0 && x0 >= 0 && y0 + wh + 6 / ww < H && xend <= W) {
        // interior window: the source pointer advances by one of two constant steps; the tail lanes of the last
        // instruction (up to 7 pixels = 6 / ww extra rows of a narrow window) still read inside the map
        const unsigned stepA = (unsigned)(dq * W + dr) * MD * 4, stepB = stepA + (unsigned)(W - ww) * MD * 4;
        const char *g = vlb + (size_t)((unsigned)(gy * W + gx) * MD) * 4;
        for (int i0 = wave_s * 8; i0 < npix; i0 += QPP) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
            gx += dr;
            const bool wrap = gx >= xend;
            gx -= wrap ? ww : 0;
            g += wrap ? stepB : stepA;
        }
    } else {
        const long zdelta = reinterpret_cast<const char *>(g_t6_zero_px + sub * 4) - vlb;
        for (int i0 = wave_s * 8; i0 < npix; i0 += QPP) {
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const long boff = inside ? (long)((size_t)((unsigned)(gy * W + gx) * MD) * 4) : zdelta;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vlb + boff),
                                             (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
            gx += dr; gy += dq;
            if (gx >= xend) { gx -= ww; ++gy; }
        }
    }
}
// Register-staged alternative (bf16 value always; fp32 value when STG == 1): 8 lanes per pixel, UN rounds of loads in flight
// before the first LDS write; bf16 is converted to fp32 on the way (same LDS image).
__device__ __forceinline__ float4_t t6_ld_px(const float *p, bool inside)
{
    const float4_t t = *reinterpret_cast<const float4_t *>(p);
    return inside ? t : (float4_t){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ float4_t t6_ld_px(const uint16_t *p, bool inside)
{
    const uint2_t r = *reinterpret_cast<const uint2_t *>(p);
    const float4_t t = {bf16lo_to_f32(r.x), bf16hi_to_f32(r.x), bf16lo_to_f32(r.y), bf16hi_to_f32(r.y)};
    return inside ? t : (float4_t){0.f, 0.f, 0.f, 0.f};
}
template <int NW, int UN, typename VT>
__device__ __forceinline__ void stage_window_regs(const VT *vl, char *dst, int y0, int x0, int wh, int ww, int H, int W,
                                                  unsigned MD, int wave_s, int lane)
{
    constexpr int QPP = NW * 8;
    const int npix = wh * ww;
    const int sub = lane & 7;
    const unsigned magic = (1u << 20) / (unsigned)ww + 1u;
    const int dq = (int)(((unsigned)QPP * magic) >> 20), dr = QPP - dq * ww;
    const int pix = wave_s * 8 + (lane >> 3);
    const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
    int gy = y0 + wy, gx = x0 + wx;
    const int xend = x0 + ww;
    const VT *vlb = vl + sub * 4;
    char *wdst = dst + (lane >> 3) * 128 + sub * 16;
    for (int i0 = wave_s * 8; i0 < npix; i0 += QPP * UN) {
        float4_t r[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const unsigned off = inside ? (unsigned)(gy * W + gx) * MD : 0u;   // (outside the map: any mapped address; the value is dropped)
            r[u] = t6_ld_px(vlb + off, inside);
            gx += dr; gy += dq;
            if (gx >= xend) { gx -= ww; ++gy; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (i0 + u * QPP < npix)   // (wave-uniform) whole 8-pixel groups; the tail lands in the window's padding
                *reinterpret_cast<float4_t *>(wdst + (size_t)(i0 + u * QPP) * 128) = r[u];
    }
}
template <int NW>
__device__ __forceinline__ void stage_window(const uint16_t *vl, char *dst, int y0, int x0, int wh, int ww, int H, int W,
                                             unsigned MD, int wave_s, int lane)
{
    stage_window_regs<NW, 4>(vl, dst, y0, x0, wh, ww, H, W, MD, wave_s, lane);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// NW waves per block, NPASS passes of NW * 16 (query, head) pairs per item, WIN arena pixels, BPC blocks per CU.
// The kernel serves exact pyramids only (checked here, from the device-side shapes: no host sync); for any other geometry it
// returns at once and the generation-4 kernel launched behind it does the work (msda_tiled6_launch).
// Phase clock ("msda_tiled" option 10): wave 0 of every block adds the shader-clock ticks it spends in each phase.
__device__ unsigned long long g_t6_prof[16];
#define T6_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// GV (gather arithmetic, for A/B): 0 v_fmac_f32_dpp (weights broadcast inside the multiply-add), 1 DPP moves + v_pk_fma_f32,
// 2 DPP moves + v_fma_f32
template <typename VT, typename OT, int NW, int NPASS, int WIN, int BPC, bool PROF = false, int GV = 0, int STG = 0>
__global__ __launch_bounds__(NW * 64, (NW * BPC + 3) / 4) void msda_fwd_tiled6_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    OT *__restrict__ out)
{
    constexpr int D = 32, PT = 4, THREADS = NW * 64, SPP = NW * 16;
    static_assert(NPASS * SPP >= 170, "an item has up to 170 queries");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *arena = smem + T6_ZPX * 128;
    int *s_box = reinterpret_cast<int *>(smem + (T6_ZPX + WIN + T6_SLACK) * 128);   // [2][4 levels][4]: min hl, min -hl, min wl, min -wl

    if (!geometry_is_pyramid(shapes, L, Lq)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned pacc[16] = {};   // (dead in the production instantiation)
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane & 3;                       // the value level this lane owns
    const unsigned MD = (unsigned)(M * D);
    const int H0 = (int)shapes[0], W0 = (int)shapes[1];
    const int HW0 = H0 * W0;
    const int ntx0 = (W0 + 15) >> 4;
    const int n_tiles = ((H0 + 7) >> 3) * ntx0;
    const unsigned n_items = (unsigned)(B * M * n_tiles);
    const int n_slots = L == 1 ? 128 : L == 2 ? 160 : L == 3 ? 168 : 170;

    // ---- per-lane constants ----
    const int quad = lane >> 2;                                  // 0..15
    const int hf = (quad >> 1) & 1;                              // which half of a pixel row this quad reads first
    const int cA = (hf * 4 + k) * 16;                            // byte offset of this lane's first 16-byte chunk (the second: ^ 64)
    const int qslot = (quad & 8) | ((0x46751320 >> ((quad & 7) * 4)) & 7);   // query slot of the quad: [0,2,3,1,5,7,6,4]
    const int kk = min(k, L - 1);                                // (lanes of levels beyond L read level L-1 and contribute nothing)
    const int Hk = H0 >> kk, Wk = W0 >> kk;
    const int v0k = (int)lsi[kk];                                // first pixel of level k in the value tensor
    // slot -> (query level, y, x, first query of the level) is the same for every item
    int sinfo[NPASS];              // level | y << 2 | x << 6 | dead << 10
    int sq0[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int s = p * SPP + wave_s * 16 + qslot;
        const int rr = min((s >= 128) + (s >= 160) + (s >= 168), 3);
        const int lo = s - (rr == 0 ? 0 : rr == 1 ? 128 : rr == 2 ? 160 : 168);
        sinfo[p] = rr | ((lo >> (4 - rr)) << 2) | ((lo & ((16 >> rr) - 1)) << 6) | ((s >= n_slots ? 1 : 0) << 10);
        sq0[p] = (rr >= 1 ? HW0 : 0) + (rr >= 2 ? HW0 >> 2 : 0) + (rr >= 3 ? HW0 >> 4 : 0);
    }

    for (int i = tid; i < T6_ZPX * 32; i += THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    if (tid < 32) s_box[tid] = T6_BIG;
    __syncthreads();

    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;

    // (b, q, m) pair index of this lane's slot in pass p of item (b, m, ty, tx) + liveness; a dead slot reads the item's
    // first query (the tile origin is always inside the map), its weights are zeroed
    auto pair_of = [&](int b, int m, int ty, int tx, int p, bool &ok) -> unsigned {
        const int sr = sinfo[p] & 3, sy = (sinfo[p] >> 2) & 15, sx = (sinfo[p] >> 6) & 15;
        const int y = ((ty * 8) >> sr) + sy, x = ((tx * 16) >> sr) + sx;
        ok = !(sinfo[p] >> 10) && y < (H0 >> sr) && x < (W0 >> sr);
        const int q = ok ? sq0[p] + y * (W0 >> sr) + x : (ty * 8) * W0 + tx * 16;
        return (unsigned)((b * Lq + q) * M + m);
    };
    float4_t lc0[NPASS], lc1[NPASS], la[NPASS];   // this lane's level: 4 x (x, y) and 4 weights per pass
    auto fetch = [&](const unsigned *pairs) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const unsigned e = (pairs[p] * (unsigned)L + (unsigned)kk) * PT;
            lc0[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2);
            lc1[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2 + 4);
            la[p] = *reinterpret_cast<const float4_t *>(attw + (size_t)e);
        }
    };
    auto decode = [&](unsigned item, int &b, int &m, int &ty, int &tx) {
        const unsigned bm = item / (unsigned)n_tiles, t = item - bm * (unsigned)n_tiles;
        const unsigned bb = bm / (unsigned)M, yy = t / (unsigned)ntx0;
        b = __builtin_amdgcn_readfirstlane((int)bb); m = __builtin_amdgcn_readfirstlane((int)(bm - bb * (unsigned)M));
        ty = __builtin_amdgcn_readfirstlane((int)yy); tx = __builtin_amdgcn_readfirstlane((int)(t - yy * (unsigned)ntx0));
    };

    // ---- item loop: the next item's locations / weights travel during the current item's gather ----
    unsigned j = blockIdx.x >> 3;
    bool have = j < ipx && xcd * ipx + j < n_items;
    int cb = 0, cm = 0, cty = 0, ctx = 0;
    unsigned pr[NPASS] = {};       // pair index of this lane's slot per pass
    bool qok[NPASS] = {};
    if (have) {
        decode(xcd * ipx + j, cb, cm, cty, ctx);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) pr[p] = pair_of(cb, cm, cty, ctx, p, qok[p]);
        fetch(pr);
    }
    int par = 0;
    while (have) {
        T6_TICK(0)   // previous item's stores, loop control
        // ---- S1: this lane's 4 points per pass (level k): weights x attention weight, corner box ----
        float w1[NPASS][4], w2[NPASS][4], w3[NPASS][4], w4[NPASS][4];
        int o[NPASS][4];               // (h_low + 1) << 16 | (w_low + 1); replaced by the LDS byte offset when level k is hot (S5)
        unsigned okm = 0;
        int r0 = T6_BIG, r1 = T6_BIG, r2 = T6_BIG, r3 = T6_BIG;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lx = i < 2 ? lc0[p][2 * i] : lc1[p][2 * i - 4], ly = i < 2 ? lc0[p][2 * i + 1] : lc1[p][2 * i - 3];
                const SamplePoint<float> sp = sample_point<float>(lx, ly, Hk, Wk);
                const bool ok = sp.ok && qok[p] && k < L;
                const float lh = sp.h_im - (float)sp.h_low, lw = sp.w_im - (float)sp.w_low;
                const float hh = 1.f - lh, hw_ = 1.f - lw;
                const float a = la[p][i];
                w1[p][i] = ok ? (hh * hw_) * a : 0.f; w2[p][i] = ok ? (hh * lw) * a : 0.f;
                w3[p][i] = ok ? (lh * hw_) * a : 0.f; w4[p][i] = ok ? (lh * lw) * a : 0.f;
                o[p][i] = ((sp.h_low + 1) << 16) | (sp.w_low + 1);
                okm |= ok ? (1u << (p * 4 + i)) : 0u;
                r0 = min(r0, ok ? sp.h_low : T6_BIG); r1 = min(r1, ok ? -sp.h_low : T6_BIG);
                r2 = min(r2, ok ? sp.w_low : T6_BIG); r3 = min(r3, ok ? -sp.w_low : T6_BIG);
            }
        }
        T6_TICK(1)   // wait for the prefetched locations + point arithmetic
        // ---- S2: boxes of all levels: lanes of equal (lane & 3) inside a row of 16, then LDS integer minima ----
        r0 = dpp_min<0x128>(dpp_min<0x124>(r0)); r1 = dpp_min<0x128>(dpp_min<0x124>(r1));   // row_ror:4, row_ror:8
        r2 = dpp_min<0x128>(dpp_min<0x124>(r2)); r3 = dpp_min<0x128>(dpp_min<0x124>(r3));
        int *boxp = s_box + par * 16;
        if ((lane & 12) == 0) {
            const unsigned a = lds_addr(boxp + k * 4);
            asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_min_i32 %0, %3 offset:8\n\tds_min_i32 %0, %4 offset:12"
                         :: "v"(a), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T6_TICK(2)   // box reduction
        __syncthreads();   // (B) boxes complete; every wave has finished gathering from the arena
        T6_TICK(3)   // barrier B
        const int4 bx = *reinterpret_cast<const int4 *>(boxp + k * 4);   // lane l < 4: the box of level l
        if (tid < 16) s_box[(par ^ 1) * 16 + tid] = T6_BIG;   // (the other buffer was read for the last time before this barrier)
        par ^= 1;
        // ---- S3: arena layout, block-uniform, kept per level in the lanes of ONE register:
        //      base | group << 16 | hot << 24 | cold << 25
        int layk;
        int ngroups;
        {
            int acc = 0, g = 0;
            int lay[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int y0 = __builtin_amdgcn_readlane(bx.x, l), ny1 = __builtin_amdgcn_readlane(bx.y, l);
                const int x0 = __builtin_amdgcn_readlane(bx.z, l), nx1 = __builtin_amdgcn_readlane(bx.w, l);
                const bool any = y0 != T6_BIG && l < L;
                const int wh = (-ny1 + 1) - y0 + 1, ww = (-nx1 + 1) - x0 + 1;     // rows y0 .. max(hl)+1
                const int np8 = any ? (wh * ww + 7) & ~7 : 0;
                const bool hot = any && np8 <= WIN && ww <= T6_ZPX - 2;
                const int need = hot ? np8 : 0;
                if (acc + need > WIN) { ++g; acc = 0; }
                lay[l] = acc | (g << 16) | (hot ? 1 << 24 : 0) | ((any && !hot) ? 1 << 25 : 0);
                acc += need;
            }
            ngroups = g + 1;
            layk = sel4(k, lay[0], lay[1], lay[2], lay[3]);
        }
        // ---- S5: one LDS byte offset per point (absolute); rejected points read the zero strip; cold levels keep (hl, wl) ----
        {
            const int y0k = bx.x, x0k = bx.z, wwk = (-bx.w + 1) - bx.z + 1;
            const int basek = layk & 0xffff;
            const bool hotk = (layk >> 24) & 1;
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hl = (o[p][i] >> 16) - 1, wl = (o[p][i] & 0xffff) - 1;
                    const bool use = (okm >> (p * 4 + i)) & 1u;
                    const int off = use ? (T6_ZPX + basek + (hl - y0k) * wwk + (wl - x0k)) * 128 : 0;
                    o[p][i] = hotk ? off : o[p][i];
                }
        }
        float acc[NPASS][8];
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[p][c] = 0.f;
        T6_TICK(4)   // arena layout, offsets

        // the next item
        j += blocks_per_xcd;
        const bool have_next = j < ipx && xcd * ipx + j < n_items;
        int nb = cb, nm = cm, nty = cty, ntx = ctx;
        unsigned npr[NPASS];
        bool nqok[NPASS];

        const VT *vb = value + ((size_t)cb * S * M + cm) * D;   // (b, pixel 0, head m)
        for (int g = 0; g < ngroups; ++g) {
            if (g > 0) __syncthreads();   // (D) the previous group has been gathered
            // ---- S4: stage the windows of this group ----
            for (int l = 0; l < L; ++l) {
                const int lay_l = __builtin_amdgcn_readlane(layk, l);
                if ((lay_l >> 16) == (g | 0x100)) {   // this group and hot
                    const int y0 = __builtin_amdgcn_readlane(bx.x, l), ny1 = __builtin_amdgcn_readlane(bx.y, l);
                    const int x0 = __builtin_amdgcn_readlane(bx.z, l), nx1 = __builtin_amdgcn_readlane(bx.w, l);
                    const int v0l = __builtin_amdgcn_readlane(v0k, l);
                    if constexpr (STG == 1)
                        stage_window_regs<NW, 6>(vb + (size_t)v0l * MD, arena + (lay_l & 0xffff) * 128, y0, x0, (-ny1 + 1) - y0 + 1,
                                                 (-nx1 + 1) - x0 + 1, H0 >> l, W0 >> l, MD, wave_s, lane);
                    else
                        stage_window<NW>(vb + (size_t)v0l * MD, arena + (lay_l & 0xffff) * 128, y0, x0, (-ny1 + 1) - y0 + 1,
                                         (-nx1 + 1) - x0 + 1, H0 >> l, W0 >> l, MD, wave_s, lane);
                }
            }
            T6_TICK(5)   // DMA issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            T6_TICK(6)   // own DMA landed
            __syncthreads();   // (C) windows complete
            T6_TICK(7)   // barrier C
            if (g == 0) {      // next item's locations / weights: in flight during the gather
                if (have_next) decode(xcd * ipx + j, nb, nm, nty, ntx);
#pragma unroll
                for (int p = 0; p < NPASS; ++p) npr[p] = pair_of(nb, nm, nty, ntx, p, nqok[p]);
                fetch(npr);
            }
            T6_TICK(8)   // next item: decode + prefetch issue
            // ---- S6: gather.  Level LQ's owner is quad lane LQ: offset and weights arrive by DPP inside the consumer ----
#define T6_HOT_POINT(P_, I_, LQ)                                                                                 \
    {                                                                                                            \
        const int b0 = qbi<LQ>(o[P_][I_]) + cA, b1 = b0 ^ 64;                                                     \
        const char *s0 = smem + b0, *s1 = smem + b1;                                                             \
        const float4_t a1 = *reinterpret_cast<const float4_t *>(s0), a2 = *reinterpret_cast<const float4_t *>(s0 + 128); \
        const float4_t a3 = *reinterpret_cast<const float4_t *>(s0 + pitch), a4 = *reinterpret_cast<const float4_t *>(s0 + pitch + 128); \
        const float4_t c1 = *reinterpret_cast<const float4_t *>(s1), c2 = *reinterpret_cast<const float4_t *>(s1 + 128); \
        const float4_t c3 = *reinterpret_cast<const float4_t *>(s1 + pitch), c4 = *reinterpret_cast<const float4_t *>(s1 + pitch + 128); \
        if constexpr (GV == 0) {                                                                                 \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
                fmac_q<LQ>(acc[P_][c], w1[P_][I_], a1[c]); fmac_q<LQ>(acc[P_][c], w2[P_][I_], a2[c]);            \
                fmac_q<LQ>(acc[P_][c], w3[P_][I_], a3[c]); fmac_q<LQ>(acc[P_][c], w4[P_][I_], a4[c]);            \
            }                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
                fmac_q<LQ>(acc[P_][4 + c], w1[P_][I_], c1[c]); fmac_q<LQ>(acc[P_][4 + c], w2[P_][I_], c2[c]);    \
                fmac_q<LQ>(acc[P_][4 + c], w3[P_][I_], c3[c]); fmac_q<LQ>(acc[P_][4 + c], w4[P_][I_], c4[c]);    \
            }                                                                                                    \
        } else {                                                                                                 \
            const float e1 = qbf<LQ>(w1[P_][I_]), e2 = qbf<LQ>(w2[P_][I_]), e3 = qbf<LQ>(w3[P_][I_]), e4 = qbf<LQ>(w4[P_][I_]); \
            if constexpr (GV == 1) {                                                                             \
                _Pragma("unroll") for (int c = 0; c < 4; c += 2) {                                               \
                    float2_t t = {acc[P_][c], acc[P_][c + 1]};                                                   \
                    t = t6_fma2(e1, (float2_t){a1[c], a1[c + 1]}, t); t = t6_fma2(e2, (float2_t){a2[c], a2[c + 1]}, t); \
                    t = t6_fma2(e3, (float2_t){a3[c], a3[c + 1]}, t); t = t6_fma2(e4, (float2_t){a4[c], a4[c + 1]}, t); \
                    acc[P_][c] = t.x; acc[P_][c + 1] = t.y;                                                      \
                    float2_t u = {acc[P_][4 + c], acc[P_][5 + c]};                                               \
                    u = t6_fma2(e1, (float2_t){c1[c], c1[c + 1]}, u); u = t6_fma2(e2, (float2_t){c2[c], c2[c + 1]}, u); \
                    u = t6_fma2(e3, (float2_t){c3[c], c3[c + 1]}, u); u = t6_fma2(e4, (float2_t){c4[c], c4[c + 1]}, u); \
                    acc[P_][4 + c] = u.x; acc[P_][5 + c] = u.y;                                                  \
                }                                                                                                \
            } else {                                                                                             \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                  \
                    acc[P_][c] = fmaf(e4, a4[c], fmaf(e3, a3[c], fmaf(e2, a2[c], fmaf(e1, a1[c], acc[P_][c])))); \
                    acc[P_][4 + c] = fmaf(e4, c4[c], fmaf(e3, c3[c], fmaf(e2, c2[c], fmaf(e1, c1[c], acc[P_][4 + c])))); \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
    }
#define T6_LEVEL(LQ)                                                                                             \
    if ((__builtin_amdgcn_readlane(layk, LQ) >> 16) == (g | 0x100)) {                                            \
        const int pitch = ((-__builtin_amdgcn_readlane(bx.w, LQ) + 1) - __builtin_amdgcn_readlane(bx.z, LQ) + 1) * 128; \
        _Pragma("unroll") for (int p = 0; p < NPASS; ++p) {                                                      \
            T6_HOT_POINT(p, 0, LQ) T6_HOT_POINT(p, 1, LQ) __builtin_amdgcn_sched_barrier(0);                     \
            T6_HOT_POINT(p, 2, LQ) T6_HOT_POINT(p, 3, LQ) __builtin_amdgcn_sched_barrier(0);                     \
        }                                                                                                        \
    }
            T6_LEVEL(0) T6_LEVEL(1) T6_LEVEL(2) T6_LEVEL(3)
            T6_TICK(9)   // gather
#undef T6_LEVEL
#undef T6_HOT_POINT
            // Cold levels of this group (window beyond the arena): the same lanes gather from global memory, one point at
            // a time, the owner lane's point data fetched by ds_bpermute (the level is a run-time value here).  A corner
            // outside the map, or of a rejected point, must not contribute whatever its clamped address holds.
            for (int l = 0; l < L; ++l) {
                const int lay_l = __builtin_amdgcn_readlane(layk, l);
                if ((lay_l >> 16) != (g | 0x200)) continue;   // this group and cold
                const int Hc = H0 >> l, Wc = W0 >> l;
                const VT *vc = vb + (size_t)__builtin_amdgcn_readlane(v0k, l) * MD;
                const int src = ((lane & ~3) | l) << 2;
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int hwp = __builtin_amdgcn_ds_bpermute(src, o[p][i]);
                        const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w1[p][i])));
                        const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w2[p][i])));
                        const float e3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w3[p][i])));
                        const float e4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w4[p][i])));
                        const int bh = (hwp >> 16) - 1, bw = (hwp & 0xffff) - 1;
                        const bool u0 = bh >= 0, u1 = bh + 1 <= Hc - 1, l0 = bw >= 0, l1 = bw + 1 <= Wc - 1;
                        const int h0 = min(max(bh, 0), Hc - 1), h1 = min(max(bh + 1, 0), Hc - 1);
                        const int c0 = min(max(bw, 0), Wc - 1), c1 = min(max(bw + 1, 0), Wc - 1);
                        const VT *p1 = vc + (size_t)((unsigned)(h0 * Wc + c0) * MD), *p2 = vc + (size_t)((unsigned)(h0 * Wc + c1) * MD);
                        const VT *p3 = vc + (size_t)((unsigned)(h1 * Wc + c0) * MD), *p4 = vc + (size_t)((unsigned)(h1 * Wc + c1) * MD);
                        const int eA = cA / 4, eB = (cA ^ 64) / 4;
                        const float4_t a1 = load4(p1 + eA), a2 = load4(p2 + eA), a3 = load4(p3 + eA), a4 = load4(p4 + eA);
                        const float4_t d1 = load4(p1 + eB), d2 = load4(p2 + eB), d3 = load4(p3 + eB), d4 = load4(p4 + eB);
                        const float f1 = (u0 && l0) ? e1 : 0.f, f2 = (u0 && l1) ? e2 : 0.f, f3 = (u1 && l0) ? e3 : 0.f, f4 = (u1 && l1) ? e4 : 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[p][c] += f1 * ((u0 && l0) ? a1[c] : 0.f) + f2 * ((u0 && l1) ? a2[c] : 0.f) +
                                         f3 * ((u1 && l0) ? a3[c] : 0.f) + f4 * ((u1 && l1) ? a4[c] : 0.f);
                            acc[p][4 + c] += f1 * ((u0 && l0) ? d1[c] : 0.f) + f2 * ((u0 && l1) ? d2[c] : 0.f) +
                                             f3 * ((u1 && l0) ? d3[c] : 0.f) + f4 * ((u1 && l1) ? d4[c] : 0.f);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (PROF) pacc[12] += 1;
            }
            T6_TICK(10)   // cold levels
            if (PROF) pacc[13] += 1;   // groups
        }
        if (PROF) pacc[14] += 1;       // items
        // ---- output: 2 x 16 bytes per lane ----
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (qok[p]) {
                OT *op = out + (size_t)pr[p] * D;
                store4(op + cA / 4, (float4_t){acc[p][0], acc[p][1], acc[p][2], acc[p][3]});
                store4(op + (cA ^ 64) / 4, (float4_t){acc[p][4], acc[p][5], acc[p][6], acc[p][7]});
            }
        }
        cb = nb; cm = nm; cty = nty; ctx = ntx; have = have_next;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) { pr[p] = npr[p]; qok[p] = nqok[p]; }
    }
    if (PROF && tid == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_t6_prof[i], (unsigned long long)pacc[i]);
    }
}

template <typename VT, typename OT, int NW, int NPASS, int WIN, int BPC, bool PROF = false, int GV = 0, int STG = 0>
int t6_go(int cus, const VT *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B,
          int S, int M, int L, int Lq, OT *out, hipStream_t st)
{
    constexpr size_t lds = (size_t)(T6_ZPX + WIN + T6_SLACK) * 128 + 128;
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled6_kernel<VT, OT, NW, NPASS, WIN, BPC, PROF, GV, STG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    VLLM_LAUNCH((msda_fwd_tiled6_kernel<VT, OT, NW, NPASS, WIN, BPC, PROF, GV, STG>), dim3((cus / 8) * 8 * BPC), dim3(NW * 64), lds, st,
                value, shapes, lsi, loc, attw, B, S, M, L, Lq, out);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled6_kernel");
    return VLLM_OK;
}

int t6_cus()
{
    const int cus = device_cus();
    return cus;
}

}  // namespace

// D == 32, P == 4, L <= 4, Lq == S, 32-bit pair / pixel offsets, 16-byte aligned tensors (checked by the caller)
bool msda_tiled6_ok(int D, int L, int P, int Lq, int S, int B, int M)
{
    return D == 32 && P == 4 && L >= 1 && L <= 4 && Lq == S && Lq >= 4096 && (long)S * M * 32 < (1L << 29) &&
           (long)B * Lq * M * L * P * 2 < (1L << 30);
}

int msda_tiled6_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, hipStream_t st)
{
    // 6 waves x 2 passes = 192 (query, head) slots for the 170 queries of an item; 560-pixel arena, 2 blocks per CU
    const int mode = msda_tiled_enabled();
    if (mode == 10) return t6_go<float, float, 6, 2, 560, 2, true>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
    if (mode == 11) return t6_go<float, float, 6, 2, 560, 2, false, 1>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
    if (mode == 12) return t6_go<float, float, 6, 2, 560, 2, false, 2>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
    if (mode == 13) return t6_go<float, float, 6, 2, 560, 2, false, 1, 1>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
    if (mode == 14) return t6_go<float, float, 6, 2, 560, 2, true, 1, 1>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
    return t6_go<float, float, 6, 2, 560, 2, false, 1>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
}

int msda_tiled6_launch_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                            const float *attw, int B, int S, int M, int L, int Lq, uint16_t *out, hipStream_t st)
{
    return t6_go<uint16_t, uint16_t, 6, 2, 560, 2, false, 1>(t6_cus(), value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st);
}

int msda6_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t6_prof), sizeof(h)) != hipSuccess) {
        set_error("msda6_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t6_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

}  // namespace vllm
