// Multi-scale deformable attention forward, LDS-tiled kernel, generation 4 ("msda_tiled" options 1 (default: 4 waves per
// block, 360-pixel windows, 3 blocks per CU), 8 (560-pixel windows, 2 blocks per CU), 2 (8 waves), 5 (phase clock)).
//
// Same tiling as msda_tiled.hip (one 8x16 query tile of one level x one head per work item, persistent blocks,
// per-level exact bounding window staged into LDS with LDS-DMA), re-organised around the instruction count, which the
// PMC profile of generation 2 showed to be the limiter (VALU 43 % + LDS 41 % busy with little overlap, 4 barriers per
// level):
//   * the window is the bounding box of ALL four corners of every accepted point, including the out-of-image ring
//     (rows -1 / H, columns -1 / W): those pixels are DMA'd from a 128-byte zero line in global memory, so the gather
//     needs ONE LDS offset per point (corners at +0, +128, +pitch, +pitch+128) and no per-corner validity logic;
//   * rejected points (and dead tile slots) aim at a zero strip in front of the window (same four relative reads);
//   * the attention weight is folded into the four bilinear weights by the lane that owns the point;
//   * sampling locations / weights go straight from global memory into the owner lane's registers, one level ahead
//     (no LDS staging, no barrier for it);
//   * the bounding box is reduced with DPP row operations + 4 LDS atomics per 16 lanes instead of 24 ds_bpermute;
//   * two barriers per level: gather(l) and point arithmetic(l+1) run in one barrier-free region, so the LDS-heavy and
//     the VALU-heavy parts of different waves overlap.
// Numerics: the integer part (sample_point) is shared with every other kernel; the weighted sum is associated as
// sum_c (w_c * a) * v_c instead of a * sum_c w_c * v_c (a few ulp; tests compare against the fp64-free oracle at 4e-6).
//
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

namespace {

constexpr int T4_TH = 8, T4_TW = 16, T4_NQ = T4_TH * T4_TW;
// Block shape: NW waves share one window.  A pass covers NW * 8 queries (8 lanes x 16 B = D 32 fp32 per query), so a
// lane serves NPASS = 16 / NW queries per item ("steps") and evaluates the points of NOWN = NPASS / 2 of them itself.
template <int NW, int TH = T4_TH>
struct T4Shape {
    static constexpr int THREADS = NW * 64, QPP = NW * 8, NPASS = TH * T4_TW / QPP, NOWN = NPASS / 2;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
};
constexpr int T4_ZPX = 48;                 // zero strip ahead of the window [pixels]; the pitch must stay <= ZPX - 2
constexpr int T4_WIN = 560;                // window budget [pixels] of the 2-blocks-per-CU configuration
constexpr int T4_WIN3 = 360;               // ... of the 3-blocks-per-CU configuration (see msda_tiled4_launch)
constexpr int T4_SLACK = 8;                // the last LDS-DMA instruction of a window may write up to 7 pixels past it
constexpr int T4_MAXL = 8;
constexpr size_t t4_lds(int win) { return (size_t)(T4_ZPX + win + T4_SLACK) * 128; }
constexpr int T4_BIG = 0x3fffffff;
struct T4Item { int b, m, q0, qW, qH, ty, tx; };   // one work item: batch, head, query tile of level-map (qH x qW) at q0

__device__ __attribute__((aligned(128))) float g_t4_zero_px[32];   // zero-initialised: DMA source of out-of-image pixels

template <int K>
__device__ __forceinline__ float qb(float x)   // value of lane K of this lane's quad
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ int qb(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ int hm(int x)   // value of the mirror lane (7 - i) of this lane's group of 8
{
    return __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);
}
__device__ __forceinline__ float hm(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float2_t fma2(float w, float2_t v, float2_t a)
{
    return __builtin_elementwise_fma((float2_t){w, w}, v, a);
}
template <int CTRL>
__device__ __forceinline__ int dpp_self(int x)
{
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ int wave_min(int v)   // v uniform inside each row of 16 -> minimum over the wave
{
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
// min over each row of 16 lanes, result in every lane of the row (VALU only)
__device__ __forceinline__ int row16_min(int v)
{
    v = min(v, dpp_self<0xB1>(v));    // quad_perm [1,0,3,2]
    v = min(v, dpp_self<0x4E>(v));    // quad_perm [2,3,0,1]
    v = min(v, dpp_self<0x141>(v));   // row_half_mirror
    v = min(v, dpp_self<0x140>(v));   // row_mirror
    return v;
}

// Phase clock (diagnostics build of the same kernel, "msda_tiled" option 5): wave 0 of every block adds the shader-clock
// ticks it spends in each phase (accumulated in LDS, added to g_t4_prof once at the end of the block);
// vllm_debug_counters() reads and clears them.
__device__ unsigned long long g_t4_prof[16];
#define T4_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// __launch_bounds__(threads, waves per SIMD): two blocks per CU
// WIN: window budget in pixels, BPC: blocks per CU the launch is sized for (LDS = (ZPX + WIN + SLACK) * 128 B per block)
// TH: query tile rows (tile = TH x 16 queries)
template <bool PROF, int NW, int WIN = T4_WIN, int BPC = 2, int TH = T4_TH>
__global__ __launch_bounds__(NW * 64, NW * BPC / 4) void msda_fwd_tiled4_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out, int skip_pyramid)
{
    constexpr int D = 32, PT = 4;
    // served by the kernel launched ahead of this one: 1 = an exact-pyramid kernel (generations 6 / 8), 2 = generation 7 (nested maps)
    if (skip_pyramid == 1 && geometry_is_pyramid(shapes, L, Lq)) return;
    if (skip_pyramid == 2 && geometry_is_nested(shapes, L, Lq)) return;
    constexpr int T4_THREADS = T4Shape<NW, TH>::THREADS, T4_QPP = T4Shape<NW, TH>::QPP, T4_NPASS = T4Shape<NW, TH>::NPASS;
    constexpr int NOWN = T4Shape<NW, TH>::NOWN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *win = reinterpret_cast<float *>(smem + T4_ZPX * 128);   // window pixel 0; the zero strip sits below it
    __shared__ int s_H[T4_MAXL], s_W[T4_MAXL], s_q0[T4_MAXL], s_tc[T4_MAXL + 1];
    __shared__ long s_v0[T4_MAXL];
    __shared__ __attribute__((aligned(16))) int s_red[NW][4];   // per wave: min hl, min -hl, min wl, min -wl
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = tid & 7;                     // 16-byte channel chunk of this lane
    unsigned pacc[16] = {};   // (dead in the production instantiation)
    // The two quads of a query's 8 lanes split the point arithmetic: quad hq owns steps 0,1 = passes 2*hq, 2*hq+1 and
    // receives the other two from its mirror lane (7 - i) with one DPP row_half_mirror per value.  A lane's step s is
    // pass (s + 2*hq) & 3; quad 0 lane k evaluates point k, quad 1 lane k point 3 - k, so that after the exchange every
    // lane holds ONE point for all four steps and the quad broadcast of lane K serves point K (quad 0) / 3 - K (quad 1).
    const int hq = (tid >> 2) & 1;
    const int kpt = hq ? 3 - (tid & 3) : (tid & 3);
    const int slot0 = tid >> 3;                  // query slot inside a pass
    const unsigned MD = (unsigned)(M * D);

    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + TH - 1) / TH) * ((W + T4_TW - 1) / T4_TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    for (int i = tid; i < T4_ZPX * 32; i += T4_THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + T4_TW - 1) / T4_TW;
    const unsigned n_items = (unsigned)(B * M * n_tiles);   // the launcher keeps every index below 2^30

    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;
    const float *zsrc = g_t4_zero_px + sub * 4;
    const char *wbase = reinterpret_cast<const char *>(win) + sub * 16;
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;

    // work item -> (batch, head, query tile); block-uniform, 32-bit arithmetic only
    auto decode = [&](unsigned item) -> T4Item {
        T4Item g;
        const unsigned bm = item / (unsigned)n_tiles, t = item - bm * (unsigned)n_tiles;
        const unsigned bb = bm / (unsigned)M;
        g.b = (int)bb; g.m = (int)(bm - bb * (unsigned)M);
        if (geo) {
            int lq = 0;
            while (lq + 1 < L && s_tc[lq + 1] <= (int)t) ++lq;
            g.qH = s_H[lq]; g.qW = s_W[lq]; g.q0 = s_q0[lq];
            const unsigned txn = (unsigned)(g.qW + T4_TW - 1) / T4_TW, tl = t - (unsigned)s_tc[lq];
            g.ty = (int)(tl / txn); g.tx = (int)(tl - (unsigned)g.ty * txn);
        } else {
            g.qH = 1; g.qW = Lq; g.q0 = 0; g.ty = 0; g.tx = (int)t;
        }
        g.b = __builtin_amdgcn_readfirstlane(g.b); g.m = __builtin_amdgcn_readfirstlane(g.m);
        g.qH = __builtin_amdgcn_readfirstlane(g.qH); g.qW = __builtin_amdgcn_readfirstlane(g.qW);
        g.q0 = __builtin_amdgcn_readfirstlane(g.q0); g.ty = __builtin_amdgcn_readfirstlane(g.ty);
        g.tx = __builtin_amdgcn_readfirstlane(g.tx);
        return g;
    };
    // (b, q, m) pair index of this lane's step (clamped to a live query) and whether the slot is live
    auto pair_of = [&](const T4Item &g, int step, bool &ok) -> unsigned {
        const int slot = ((step + NOWN * hq) & (T4_NPASS - 1)) * T4_QPP + slot0;
        const int y = g.ty * TH + slot / T4_TW, x = g.tx * T4_TW + slot % T4_TW;
        ok = y < g.qH && x < g.qW;
        const int q = g.q0 + (ok ? y : 0) * g.qW + (ok ? x : 0);
        return (unsigned)((g.b * Lq + q) * M + g.m);
    };

    // The locations / weights of an item's first level are requested during the previous item's last level (the
    // phase clock showed that wait as a quarter of the kernel), so the item state is loop-carried.
    unsigned j = blockIdx.x >> 3;
    bool have = j < ipx && xcd * ipx + j < n_items;
    T4Item cur = {0, 0, 0, 1, 1, 0, 0};
    unsigned q01[NOWN] = {};   // pair index of the steps this lane evaluates points for
    bool qok[NOWN] = {};
    float2_t lc[NOWN] = {};
    float la[NOWN] = {};
    if (have) {
        cur = decode(xcd * ipx + j);
#pragma unroll
        for (int p = 0; p < NOWN; ++p) {
            q01[p] = pair_of(cur, p, qok[p]);
            lc[p] = *reinterpret_cast<const float2_t *>(loc + (size_t)((q01[p] * L * PT + kpt) * 2));
            la[p] = attw[(size_t)(q01[p] * L * PT + kpt)];
        }
    }
    long nshH = shapes[0], nshW = shapes[1], nlsi = lsi[0];   // level constants, fetched one level ahead
    while (have) {
        const int m = cur.m;
        const long b = cur.b;
        j += blocks_per_xcd;
        const bool have_next = j < ipx && xcd * ipx + j < n_items;
        // the next item (this one again when there is none: its loads are then simply unused)
        const T4Item nxt = have_next ? decode(xcd * ipx + j) : cur;
        unsigned nq01[NOWN];
        bool nqok[NOWN];
#pragma unroll
        for (int p = 0; p < NOWN; ++p) nq01[p] = pair_of(nxt, p, nqok[p]);
        T4_TICK(12)   // next item's decode
        float2_t acc2[T4_NPASS][2];   // this lane's 4 channels of each step's query, as two 2-wide halves
#pragma unroll
        for (int p = 0; p < T4_NPASS; ++p) acc2[p][0] = acc2[p][1] = (float2_t){0.f, 0.f};

        for (int l = 0; l < L; ++l) {
            // Level constants: scalar loads issued ONE LEVEL AHEAD (a memory round trip at the top of the level, LDS or
            // scalar cache alike, cost 800-1500 cycles on the phase clock with every wave of the block waiting on it).
            const int H = (int)nshH, W = (int)nshW;
            const float *vl = value + (b * (long)S + nlsi) * MD + (long)m * D + sub * 4;
            {
                const int ln = l + 1 < L ? l + 1 : 0;
                nshH = shapes[2 * ln]; nshW = shapes[2 * ln + 1]; nlsi = lsi[ln];
            }

            T4_TICK(11)   // level constants from LDS
            // ---- A: this lane's point of its two own steps; weights (x attention weight); corner bounding box ----
            int hl[T4_NPASS], wl[T4_NPASS];
            float w1[T4_NPASS], w2[T4_NPASS], w3[T4_NPASS], w4[T4_NPASS];
            unsigned okmask = 0;
            int r0 = T4_BIG, r1 = T4_BIG, r2 = T4_BIG, r3 = T4_BIG;   // min hl, min -hl, min wl, min -wl
#pragma unroll
            for (int p = 0; p < NOWN; ++p) {
                const SamplePoint<float> sp = sample_point<float>(lc[p].x, lc[p].y, H, W);
                const bool ok = sp.ok && qok[p] && H > 0 && W > 0;   // (empty level: no corner inside, adds nothing)
                hl[p] = sp.h_low; wl[p] = sp.w_low;
                const float lh = sp.h_im - (float)sp.h_low, lw = sp.w_im - (float)sp.w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float a = la[p];
                w1[p] = ok ? (hh * hw) * a : 0.f; w2[p] = ok ? (hh * lw) * a : 0.f;
                w3[p] = ok ? (lh * hw) * a : 0.f; w4[p] = ok ? (lh * lw) * a : 0.f;
                okmask |= ok ? (1u << p) : 0u;
                r0 = min(r0, ok ? sp.h_low : T4_BIG); r1 = min(r1, ok ? -sp.h_low : T4_BIG);
                r2 = min(r2, ok ? sp.w_low : T4_BIG); r3 = min(r3, ok ? -sp.w_low : T4_BIG);
            }
            T4_TICK(9)   // point arithmetic
            // wave minimum: DPP inside each row of 16, the four rows through SGPRs; lane 0 publishes the wave's box
            r0 = wave_min(row16_min(r0)); r1 = wave_min(row16_min(r1));
            r2 = wave_min(row16_min(r2)); r3 = wave_min(row16_min(r3));
            if (lane == 0) *reinterpret_cast<int4 *>(&s_red[wave_s][0]) = make_int4(r0, r1, r2, r3);
            T4_TICK(10)   // box reduction
            // The next level's (behind the last level: the next item's first level's) locations / weights travel while
            // this level is staged and gathered.  ONE unconditional load site: with the loads in two branches the
            // compiler merged them through register copies and waited for the data right here, every level.
            {
                const bool lastl = l + 1 == L;
                const unsigned lv = lastl ? 0u : (unsigned)(l + 1);
#pragma unroll
                for (int p = 0; p < NOWN; ++p) {
                    const unsigned e = ((lastl ? nq01[p] : q01[p]) * L + lv) * PT + kpt;
                    lc[p] = *reinterpret_cast<const float2_t *>(loc + (size_t)(e * 2));
                    la[p] = attw[(size_t)e];
                }
            }
            T4_TICK(1)   // prefetch issue
            __syncthreads();   // (B) box complete; every wave has finished gathering the previous window
            T4_TICK(2)   // barrier B
            // (the next writes to s_red happen behind barrier (C), which every path below executes)
            // lane i reads wave (i mod NW)'s box; a row of 16 lanes then holds every wave's at least once
            const int4 bw = *reinterpret_cast<const int4 *>(&s_red[lane & (NW - 1)][0]);
            const int y0 = __builtin_amdgcn_readfirstlane(row16_min(bw.x)), ny1 = __builtin_amdgcn_readfirstlane(row16_min(bw.y));
            const int x0 = __builtin_amdgcn_readfirstlane(row16_min(bw.z)), nx1 = __builtin_amdgcn_readfirstlane(row16_min(bw.w));
            const int wh = (-ny1 + 1) - y0 + 1, ww = (-nx1 + 1) - x0 + 1;   // rows y0 .. max(hl)+1, columns x0 .. max(wl)+1
            const int npix = wh * ww;
            if (__builtin_expect(y0 == T4_BIG || npix > WIN || ww > T4_ZPX - 2, 0)) {
                // Cold (block-uniform): no accepted point at this level, or a window beyond the LDS budget.  The latter
                // gathers the level from global memory in a compact ROLLED loop (locations / weights re-read from
                // global, they are L2-hot): unrolled, this path was 1300 instructions in the middle of the hot loop.
                if (PROF) { pacc[y0 != T4_BIG ? 13 : 14] += 1; }
                if (y0 != T4_BIG) {
                    okmask |= (unsigned)hm((int)okmask) << NOWN;
#pragma unroll
                    for (int p = 0; p < NOWN; ++p) {
                        hl[NOWN + p] = hm(hl[p]); wl[NOWN + p] = hm(wl[p]);
                        w1[NOWN + p] = hm(w1[p]); w2[NOWN + p] = hm(w2[p]); w3[NOWN + p] = hm(w3[p]); w4[NOWN + p] = hm(w4[p]);
                    }
                    // one step (16 corner loads per lane) at a time; more loads in flight cost more registers than the
                    // compiler handles gracefully here.  A corner outside the map, or of a rejected point, must not
                    // contribute whatever its clamped address holds.
#pragma unroll
                    for (int p = 0; p < T4_NPASS; ++p) {
#define T4_GPOINT(K)                                                                                             \
    {                                                                                                            \
        const int bh = qb<K>(hl[p]), bx = qb<K>(wl[p]);                                                          \
        const float b1 = qb<K>(w1[p]), b2 = qb<K>(w2[p]), b3 = qb<K>(w3[p]), b4 = qb<K>(w4[p]);                  \
        const bool bo = qb<K>((int)((okmask >> p) & 1u)) != 0;                                                   \
        const bool u0 = bo && bh >= 0, u1 = bo && bh + 1 <= H - 1, l0 = bx >= 0, l1 = bx + 1 <= W - 1;           \
        const int h0 = min(max(bh, 0), H - 1), h1 = min(max(bh + 1, 0), H - 1);                                  \
        const int c0 = min(max(bx, 0), W - 1), c1 = min(max(bx + 1, 0), W - 1);                                  \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h0 * W + c0) * MD));    \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h0 * W + c1) * MD));    \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h1 * W + c0) * MD));    \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h1 * W + c1) * MD));    \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
            acc2[p][c >> 1][c & 1] += b1 * ((u0 && l0) ? v1[c] : 0.f) + b2 * ((u0 && l1) ? v2[c] : 0.f) +        \
                                      b3 * ((u1 && l0) ? v3[c] : 0.f) + b4 * ((u1 && l1) ? v4[c] : 0.f);         \
        }                                                                                                        \
    }
                        T4_GPOINT(0) T4_GPOINT(1) T4_GPOINT(2) T4_GPOINT(3)
#undef T4_GPOINT
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                T4_TICK(15)   // cold path
                continue;
            }
#pragma unroll
            for (int p = 0; p < NOWN; ++p) {            // the other half of the steps comes from the mirror lane's own steps
                w1[NOWN + p] = hm(w1[p]); w2[NOWN + p] = hm(w2[p]); w3[NOWN + p] = hm(w3[p]); w4[NOWN + p] = hm(w4[p]);
            }

            T4_TICK(3)   // box read, weight exchange
            // ---- B: stage the window (LDS-DMA, 8 pixels of 128 B per wave instruction); ring pixels come from the zero line
            {
                const unsigned magic = (1u << 20) / (unsigned)ww + 1u;      // pix / ww for pix * ww < 2^20
                const int dq = (int)(((unsigned)T4_QPP * magic) >> 20), dr = T4_QPP - dq * ww;   // QPP pixels per round of the block
                const int pix = wave_s * 8 + (lane >> 3);
                const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
                int gy = y0 + wy, gx = x0 + wx;
                const int xend = x0 + ww;
                const char *vlb = reinterpret_cast<const char *>(vl);
                if (y0 >= 0 && x0 >= 0 && y0 + wh + 6 / ww < H && xend <= W) {
                    // interior window: the source pointer advances by one of two constant steps.  The lanes behind the
                    // window's last pixel (tail of the last instruction, up to 7 pixels = 6 / ww extra rows of a narrow
                    // window; their copies land in the slack) still read inside the map: windows that come closer to the
                    // last row than that take the general loop.
                    const unsigned stepA = (unsigned)(dq * W + dr) * MD * 4, stepB = stepA + (unsigned)(W - ww) * MD * 4;
                    const char *g = vlb + (size_t)((unsigned)(gy * W + gx) * MD) * 4;
                    for (int i0 = wave_s * 8; i0 < npix; i0 += T4_QPP) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
                        gx += dr;
                        const bool wrap = gx >= xend;
                        gx -= wrap ? ww : 0;
                        g += wrap ? stepB : stepA;
                    }
                } else {
                    const long zdelta = reinterpret_cast<const char *>(zsrc) - vlb;
                    for (int i0 = wave_s * 8; i0 < npix; i0 += T4_QPP) {
                        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        const long boff = inside ? (long)((size_t)((unsigned)(gy * W + gx) * MD) * 4) : zdelta;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vlb + boff),
                                                         (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
                        gx += dr; gy += dq;
                        if (gx >= xend) { gx -= ww; ++gy; }
                    }
                }
            }
            T4_TICK(4)   // DMA issue loop
            // one LDS byte offset per point; a rejected point reads the zero strip
            const int pitchB = ww * 128;
            int o[T4_NPASS];
#pragma unroll
            for (int p = 0; p < NOWN; ++p) {
                o[p] = ((okmask >> p) & 1u) ? ((hl[p] - y0) * ww + (wl[p] - x0)) * 128 : -T4_ZPX * 128;
                o[NOWN + p] = hm(o[p]);
            }
            T4_TICK(5)   // offsets
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            T4_TICK(6)   // own DMA + prefetch complete
            __syncthreads();   // (C) window complete
            T4_TICK(7)   // barrier C

            // ---- C: gather from LDS; the quad's lane K broadcasts point K's offset / weights (DPP) ----
            const char *wrow1 = wbase + pitchB;
            // (explicit 2-wide fma = v_pk_fma_f32 with the weight broadcast by op_sel: left to itself the SLP vectoriser pairs
            // channels of DIFFERENT steps and pays for it in register shuffles.  Reading a whole step ahead of the sums
            // was measured and does not help: the phase is bound by LDS bank conflicts (two of the four 64-byte segments
            // of a 16-lane group share a bank half whenever their pixels have equal parity), not by LDS latency.)
#define T4_POINT(K)                                                                                              \
    {                                                                                                            \
        const int off = qb<K>(o[p]);                                                                             \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(wbase + off);                                    \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(wbase + off + 128);                              \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(wrow1 + off);                                    \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(wrow1 + off + 128);                              \
        const float b1 = qb<K>(w1[p]), b2 = qb<K>(w2[p]), b3 = qb<K>(w3[p]), b4 = qb<K>(w4[p]);                  \
        acc2[p][0] = fma2(b4, v4.lo, fma2(b3, v3.lo, fma2(b2, v2.lo, fma2(b1, v1.lo, acc2[p][0]))));              \
        acc2[p][1] = fma2(b4, v4.hi, fma2(b3, v3.hi, fma2(b2, v2.hi, fma2(b1, v1.hi, acc2[p][1]))));              \
    }
#pragma unroll
            for (int p = 0; p < T4_NPASS; ++p) {
                T4_POINT(0) T4_POINT(1) T4_POINT(2) T4_POINT(3)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef T4_POINT
            T4_TICK(8)   // gather
        }
#pragma unroll
        for (int p = 0; p < T4_NPASS; ++p) {
            bool ok;
            const unsigned qi = pair_of(cur, p, ok);
            if (ok) {
                float4_t ov = {acc2[p][0].x, acc2[p][0].y, acc2[p][1].x, acc2[p][1].y};
                *reinterpret_cast<float4_t *>(out + (size_t)qi * D + sub * 4) = ov;
            }
        }
        cur = nxt; have = have_next;
#pragma unroll
        for (int p = 0; p < NOWN; ++p) { q01[p] = nq01[p]; qok[p] = nqok[p]; }
        T4_TICK(0)   // item epilogue: stores
    }
    if (PROF && tid == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_t4_prof[i], (unsigned long long)pacc[i]);
    }
}

}  // namespace

int msda_tiled4_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                       const float *attw, int B, int S, int M, int L, int Lq, float *out, int skip_pyramid,
                       hipStream_t st)
{
    const int cus = device_cus();
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled4_kernel<false, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)t4_lds(T4_WIN));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled4_kernel<false, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)t4_lds(T4_WIN));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled4_kernel<true, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)t4_lds(T4_WIN));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled4_kernel<false, 4, T4_WIN3, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)t4_lds(T4_WIN3));
    }
    const int mode = msda_tiled_enabled();
#define T4_GO(PROF, NW, WIN, BPC)                                                                                     \
    VLLM_LAUNCH((msda_fwd_tiled4_kernel<PROF, NW, WIN, BPC>), dim3((cus / 8) * 8 * BPC), dim3(NW * 64), t4_lds(WIN), st, value, \
                shapes, lsi, loc, attw, B, S, M, L, Lq, out, skip_pyramid)
    // Window sizes at the cfg-4 encoder shape: median 108, mean 205, 90th percentile 308 pixels; 7.7 % of the (tile, level)
    // pairs exceed 360 against 6.9 % that exceed 560 (coarse query level -> fine value level either way).  A 360-pixel
    // budget is 52 KiB of LDS per block = THREE blocks per CU instead of two for 0.9 % more cold pairs: 582 vs 612 us.
    // (256-pixel windows and FOUR blocks per CU: 727 us -- 14 % cold pairs cost more than the fourth block hides.  Other
    // tile shapes through the TH template parameter: 16x16 queries / 560 pixels / 2 blocks 651 us, 4x16 queries / 256
    // pixels / 4 blocks 587 us against 594 us for the default on the same box -- inside the noise, not adopted.)
    if (mode == 5) T4_GO(true, 4, T4_WIN, 2);          // phase clock (diagnostics)
    else if (mode == 2) T4_GO(false, 8, T4_WIN, 2);    // 8 waves per block, 2 blocks per CU
    else if (mode == 8) T4_GO(false, 4, T4_WIN, 2);    // 4 waves per block, 560-pixel windows, 2 blocks per CU (608 us)
    else T4_GO(false, 4, T4_WIN3, 3);                  // 360-pixel windows, 3 blocks per CU (583 us, same box)
#undef T4_GO
    VLLM_CHECK_LAUNCH("msda_fwd_tiled4_kernel");
    return VLLM_OK;
}

int msda_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t4_prof), sizeof(h)) != hipSuccess) {
        set_error("msda_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t4_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

}  // namespace vllm
