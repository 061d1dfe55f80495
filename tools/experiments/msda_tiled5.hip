// Multi-scale deformable attention forward, LDS-tiled kernel, generation 5: producer / consumer wave specialisation
// ("msda_tiled" option 6).
//
// Generation 4 (msda_tiled4.hip) runs every phase of a (query tile, level) step in every wave: point arithmetic, bounding
// box, LDS-DMA issue (13.5 % of the kernel on its phase clock), wait for the window (10.5 %), gather (28.6 %).  The phases
// of one block are serial; only the second block of the CU overlaps them, by chance.  Here ONE block per CU holds two
// windows and two kinds of waves:
//   * NP producer waves stage the window of step s+1 (bounding box of all 128 x 4 sampling points -- every producer wave
//     evaluates the whole box itself, so producers never synchronise with each other -- then the LDS-DMA of their share of
//     the window rows, then s_waitcnt vmcnt(0));
//   * 8 consumer waves gather step s from the other window (generation 4's gather, point arithmetic and cold path,
//     unchanged arithmetic: the two generations are bit-identical).
// One s_barrier per step, executed by both kinds (n_steps barriers in each role): barrier(s) publishes window s and its box
// (sbox[s & 1]) and retires every read of the window that step s+1 overwrites.  No spin loops, no flags.
//   consumers:  [points(s)]  barrier(s)  [gather(s)]  [points(s+1)]  barrier(s+1) ...
//   producers:  [stage(0)]   barrier(0)  [stage(1)]                  barrier(1)   ...
// STATUS (round 1): correct (bit-identical to generation 4, same tests) but NOT the default: 743 us (4 producer waves) /
// 837 us (8) against 613-640 us for generation 4 at the cfg-4 encoder shape.  Timing-only ablations of this kernel: without
// the producers' box arithmetic -175 us, without the LDS-DMA -154 us, without the consumers' gather -79 us; what is left
// (~340 us = 2500 cycles per step, 320 steps per CU) is one global-memory round trip per interval: with two windows the
// DMA of step k can only be issued after barrier(k-1) and must have landed before barrier(k), so every interval contains a
// full L2 / HBM latency that nothing overlaps (issuing the DMA before the next step's box arithmetic instead of after it
// changed nothing: 764 us).  Generation 4 hides the same latency behind the SECOND block of the CU.  Beating it needs a
// window ring deeper than two (smaller tiles / windows), not role specialisation on two windows.
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

namespace {

constexpr int T5_TH = 8, T5_TW = 16, T5_NQ = T5_TH * T5_TW;
constexpr int T5_NC = 8;                       // consumer waves
constexpr int T5_CTHREADS = T5_NC * 64;        // 512 consumer lanes: 8 lanes x 16 B per query, 64 queries per pass
constexpr int T5_QPP = T5_NC * 8, T5_NPASS = T5_NQ / T5_QPP;   // 2 passes; a lane evaluates the points of ONE of them
constexpr int T5_ZPX = 48;                     // zero strip [pixels]; the window pitch must stay <= ZPX - 2
constexpr int T5_WIN = 560;                    // window budget [pixels]
constexpr int T5_WINPIX = T5_WIN + 8;          // + LDS-DMA slack
constexpr int T5_MAXL = 8;
constexpr size_t T5_LDS = (size_t)(T5_ZPX + 2 * T5_WINPIX) * 128;
constexpr int T5_BIG = 0x3fffffff;
struct T5Item { int b, m, q0, qW, qH, ty, tx; };

__device__ __attribute__((aligned(128))) float g_t5_zero_px[32];   // DMA source of out-of-image pixels

template <int K> __device__ __forceinline__ float qb5(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ int qb5(int x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, true); }
__device__ __forceinline__ int hm5(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true); }
__device__ __forceinline__ float hm5(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float2_t fma2_5(float w, float2_t v, float2_t a) { return __builtin_elementwise_fma((float2_t){w, w}, v, a); }
template <int CTRL> __device__ __forceinline__ int dpp_self5(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_min5(int v)
{
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int row16_min5(int v)
{
    v = min(v, dpp_self5<0xB1>(v));
    v = min(v, dpp_self5<0x4E>(v));
    v = min(v, dpp_self5<0x141>(v));
    v = min(v, dpp_self5<0x140>(v));
    return v;
}

template <int NP>
__global__ __launch_bounds__((T5_NC + NP) * 64, 1) void msda_fwd_tiled5_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out)
{
    constexpr int D = 32, PT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [zero strip | window 0 | window 1]
    __shared__ int s_H[T5_MAXL], s_W[T5_MAXL], s_q0[T5_MAXL], s_tc[T5_MAXL + 1];
    __shared__ long s_v0[T5_MAXL];
    __shared__ __attribute__((aligned(16))) int s_box[2][4];   // per window buffer: y0, x0, ww, state (0 empty, 1 staged, 2 too large)
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave_s >= T5_NC;
    const unsigned MD = (unsigned)(M * D);

    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + T5_TH - 1) / T5_TH) * ((W + T5_TW - 1) / T5_TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    for (int i = tid; i < T5_ZPX * 32; i += (T5_NC + NP) * 64) reinterpret_cast<float *>(smem)[i] = 0.f;
    __syncthreads();
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + T5_TW - 1) / T5_TW;
    const unsigned n_items = (unsigned)(B * M * n_tiles);   // the launcher keeps every index below 2^30
    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;

    auto decode = [&](unsigned item) -> T5Item {
        T5Item g;
        const unsigned bm = item / (unsigned)n_tiles, t = item - bm * (unsigned)n_tiles;
        const unsigned bb = bm / (unsigned)M;
        g.b = (int)bb; g.m = (int)(bm - bb * (unsigned)M);
        if (geo) {
            int lq = 0;
            while (lq + 1 < L && s_tc[lq + 1] <= (int)t) ++lq;
            g.qH = s_H[lq]; g.qW = s_W[lq]; g.q0 = s_q0[lq];
            const unsigned txn = (unsigned)(g.qW + T5_TW - 1) / T5_TW, tl = t - (unsigned)s_tc[lq];
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
    // (b, q, m) pair index of tile slot `slot` (clamped to a live query) and whether the slot is live
    auto pair_of_slot = [&](const T5Item &g, int slot, bool &ok) -> unsigned {
        const int y = g.ty * T5_TH + slot / T5_TW, x = g.tx * T5_TW + slot % T5_TW;
        ok = y < g.qH && x < g.qW;
        const int q = g.q0 + (ok ? y : 0) * g.qW + (ok ? x : 0);
        return (unsigned)((g.b * Lq + q) * M + g.m);
    };

    unsigned j = blockIdx.x >> 3;
    bool have = j < ipx && xcd * ipx + j < n_items;

    if (producer) {
        // =========================== producers: window of step k staged during interval k-1 ===========================
        // Inside an interval the order is: location loads of step k+2, LDS-DMA of step k (its box was computed one interval
        // earlier), box of step k+1 (VALU, under the DMA's flight time), s_waitcnt vmcnt(0), barrier(k).  The DMA latency
        // (~1.5 us to the last byte) would otherwise sit between the box and the barrier of every step.
        const int pw = wave_s - T5_NC;   // producer wave index (SGPR)
        // lane i looks at tile slots 2i, 2i+1 (all four points of each): every producer wave covers the whole tile
        struct Pos { unsigned j; T5Item g; int l; bool valid; };
        auto advance = [&](const Pos &p) -> Pos {
            Pos n = p;
            if (!p.valid) return n;
            if (p.l + 1 < L) { n.l = p.l + 1; return n; }
            n.j = p.j + blocks_per_xcd;
            n.l = 0;
            n.valid = n.j < ipx && xcd * ipx + n.j < n_items;
            if (n.valid) n.g = decode(xcd * ipx + n.j);
            return n;
        };
        struct Locs { float4_t v[2][2]; bool ok[2]; };
        auto load_locs = [&](const Pos &p) -> Locs {   // ONE load site (clamped to a valid step: unused data otherwise)
            Locs r;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned pq = pair_of_slot(p.g, 2 * lane + i, r.ok[i]);
                r.ok[i] = r.ok[i] && p.valid;
                const float *lp = loc + (size_t)((pq * L + (unsigned)p.l) * PT) * 2;
                r.v[i][0] = *reinterpret_cast<const float4_t *>(lp);
                r.v[i][1] = *reinterpret_cast<const float4_t *>(lp + 4);
            }
            return r;
        };
        struct Box { int y0, ny1, x0, nx1; };
        auto box_of = [&](const Locs &c, const Pos &p) -> Box {
            const int H = s_H[p.l], W = s_W[p.l];
            int r0 = T5_BIG, r1 = T5_BIG, r2 = T5_BIG, r3 = T5_BIG;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    const float lx = c.v[i][k >> 1][(k & 1) * 2], ly = c.v[i][k >> 1][(k & 1) * 2 + 1];
                    const SamplePoint<float> sp = sample_point<float>(lx, ly, H, W);
                    const bool ok = sp.ok && c.ok[i];
                    r0 = min(r0, ok ? sp.h_low : T5_BIG); r1 = min(r1, ok ? -sp.h_low : T5_BIG);
                    r2 = min(r2, ok ? sp.w_low : T5_BIG); r3 = min(r3, ok ? -sp.w_low : T5_BIG);
                }
            Box bx;
            bx.y0 = wave_min5(row16_min5(r0)); bx.ny1 = wave_min5(row16_min5(r1));
            bx.x0 = wave_min5(row16_min5(r2)); bx.nx1 = wave_min5(row16_min5(r3));
            return bx;
        };
        Pos pd;   // the step staged in the current interval
        pd.j = j; pd.l = 0; pd.valid = have; pd.g = (T5Item){0, 0, 0, 1, 1, 0, 0};
        if (have) pd.g = decode(xcd * ipx + j);
        Pos pb = advance(pd);     // the step whose box is computed in the current interval
        Pos pn = advance(pb);     // the step whose locations are requested in the current interval
        Box bd = {T5_BIG, T5_BIG, T5_BIG, T5_BIG};
        Locs lb = load_locs(pd);
        if (pd.valid) bd = box_of(lb, pd);
        lb = load_locs(pb);
        int buf = 0;
        while (pd.valid) {
            const Locs ln = load_locs(pn);   // in flight behind the DMA below
            const int l = pd.l, H = s_H[l], W = s_W[l];
            const char *vlb = reinterpret_cast<const char *>(value + ((long)pd.g.b * S + s_v0[l]) * MD + (long)pd.g.m * D + (lane & 7) * 4);
            const int y0 = bd.y0, x0 = bd.x0;
            const int wh = (-bd.ny1 + 1) - y0 + 1, ww = (-bd.nx1 + 1) - x0 + 1;
            const int npix = wh * ww;
            const int state = y0 == T5_BIG ? 0 : ((npix > T5_WIN || ww > T5_ZPX - 2) ? 2 : 1);
            if (pw == 0 && lane == 0) *reinterpret_cast<int4 *>(&s_box[buf][0]) = make_int4(y0, x0, ww, state);
            if (state == 1) {
                float *win = reinterpret_cast<float *>(smem + (size_t)(T5_ZPX + buf * T5_WINPIX) * 128);
                constexpr int PPR = NP * 8;   // pixels per round of the producer group
                const unsigned magic = (1u << 20) / (unsigned)ww + 1u;      // pix / ww for pix * ww < 2^20
                const int dq = (int)(((unsigned)PPR * magic) >> 20), dr = PPR - dq * ww;
                const int pix = pw * 8 + (lane >> 3);
                const int wy = (int)(((unsigned)pix * magic) >> 20), wx = pix - wy * ww;
                int gy = y0 + wy, gx = x0 + wx;
                const int xend = x0 + ww;
                if (y0 >= 0 && x0 >= 0 && y0 + wh + 6 / ww < H && xend <= W) {
                    const unsigned stepA = (unsigned)(dq * W + dr) * MD * 4, stepB = stepA + (unsigned)(W - ww) * MD * 4;
                    const char *g = vlb + (size_t)((unsigned)(gy * W + gx) * MD) * 4;
                    for (int i0 = pw * 8; i0 < npix; i0 += PPR) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
                        gx += dr;
                        const bool wrap = gx >= xend;
                        gx -= wrap ? ww : 0;
                        g += wrap ? stepB : stepA;
                    }
                } else {
                    const long zdelta = reinterpret_cast<const char *>(g_t5_zero_px + (lane & 7) * 4) - vlb;
                    for (int i0 = pw * 8; i0 < npix; i0 += PPR) {
                        const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        const long boff = inside ? (long)((size_t)((unsigned)(gy * W + gx) * MD) * 4) : zdelta;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vlb + boff),
                                                         (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
                        gx += dr; gy += dq;
                        if (gx >= xend) { gx -= ww; ++gy; }
                    }
                }
            }
            // box of the NEXT step while this step's window is in flight
            Box bn = {T5_BIG, T5_BIG, T5_BIG, T5_BIG};
            if (pb.valid) bn = box_of(lb, pb);
            // own DMA pieces (and the location prefetch) have landed, the box store has retired; barrier(k) publishes this
            // step's window + box.  One asm so that no memory operation is moved across it by the compiler.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf ^= 1;
            pd = pb; pb = pn; pn = advance(pn);
            bd = bn; lb = ln;
        }
        return;
    }

    // ================================ consumers: generation 4's point arithmetic + gather ================================
    const int sub = tid & 7;                     // 16-byte channel chunk of this lane
    // The two quads of a query's 8 lanes split the passes: quad hq evaluates the points of pass hq and receives the other
    // pass's from its mirror lane (7 - i); quad 0 lane k evaluates point k, quad 1 lane k point 3 - k.
    const int hq = (tid >> 2) & 1;
    const int kpt = hq ? 3 - (tid & 3) : (tid & 3);
    const int slot0 = tid >> 3;                  // query slot inside a pass (0..63)
    const char *wbase = smem + sub * 16;         // absolute LDS addressing: offsets include the window base
    auto pair_of = [&](const T5Item &g, int step, bool &ok) -> unsigned {
        const int slot = ((step + hq) & (T5_NPASS - 1)) * T5_QPP + slot0;
        return pair_of_slot(g, slot, ok);
    };
    T5Item cur = {0, 0, 0, 1, 1, 0, 0};
    unsigned q01 = 0u;
    bool qok = false;
    float2_t lc = {0.f, 0.f};
    float la = 0.f;
    if (have) {
        cur = decode(xcd * ipx + j);
        q01 = pair_of(cur, 0, qok);
        lc = *reinterpret_cast<const float2_t *>(loc + (size_t)((q01 * L * PT + kpt) * 2));
        la = attw[(size_t)(q01 * L * PT + kpt)];
    }
    int buf = 0;
    while (have) {
        const int m = cur.m;
        const long b = cur.b;
        j += blocks_per_xcd;
        const bool have_next = j < ipx && xcd * ipx + j < n_items;
        const T5Item nxt = have_next ? decode(xcd * ipx + j) : cur;
        bool nqok;
        const unsigned nq01 = pair_of(nxt, 0, nqok);
        float2_t acc2[T5_NPASS][2];
#pragma unroll
        for (int p = 0; p < T5_NPASS; ++p) acc2[p][0] = acc2[p][1] = (float2_t){0.f, 0.f};

        for (int l = 0; l < L; ++l) {
            const int H = s_H[l], W = s_W[l];
            const float *vl = value + (b * (long)S + s_v0[l]) * MD + (long)m * D + sub * 4;
            // ---- this lane's point of its own pass; weights (x attention weight) ----
            int hl[T5_NPASS], wl[T5_NPASS];
            float w1[T5_NPASS], w2[T5_NPASS], w3[T5_NPASS], w4[T5_NPASS];
            unsigned okmask = 0;
            {
                const SamplePoint<float> sp = sample_point<float>(lc.x, lc.y, H, W);
                const bool ok = sp.ok && qok;
                hl[0] = sp.h_low; wl[0] = sp.w_low;
                const float lh = sp.h_im - (float)sp.h_low, lw = sp.w_im - (float)sp.w_low;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float a = la;
                w1[0] = ok ? (hh * hw) * a : 0.f; w2[0] = ok ? (hh * lw) * a : 0.f;
                w3[0] = ok ? (lh * hw) * a : 0.f; w4[0] = ok ? (lh * lw) * a : 0.f;
                okmask = ok ? 1u : 0u;
            }
            // next level's (behind the last level: the next item's first level's) location / weight: ONE load site
            {
                const bool lastl = l + 1 == L;
                const unsigned lv = lastl ? 0u : (unsigned)(l + 1);
                const unsigned e = ((lastl ? nq01 : q01) * L + lv) * PT + kpt;
                lc = *reinterpret_cast<const float2_t *>(loc + (size_t)(e * 2));
                la = attw[(size_t)e];
            }
            // barrier(s): the producers have published this step's window and box; every LDS read of the previous gather has
            // returned (the window it used is overwritten after this barrier).  The location prefetch stays in flight.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int4 bx = *reinterpret_cast<const int4 *>(&s_box[buf][0]);
            const int y0 = __builtin_amdgcn_readfirstlane(bx.x), x0 = __builtin_amdgcn_readfirstlane(bx.y);
            const int ww = __builtin_amdgcn_readfirstlane(bx.z), state = __builtin_amdgcn_readfirstlane(bx.w);
            const int wbuf = buf;
            buf ^= 1;
            if (__builtin_expect(state != 1, 0)) {
                // cold (block-uniform): no accepted point at this level, or a window beyond the LDS budget -> global gather
                if (state == 2) {
                    okmask |= (unsigned)hm5((int)okmask) << 1;
                    hl[1] = hm5(hl[0]); wl[1] = hm5(wl[0]);
                    w1[1] = hm5(w1[0]); w2[1] = hm5(w2[0]); w3[1] = hm5(w3[0]); w4[1] = hm5(w4[0]);
#pragma unroll
                    for (int p = 0; p < T5_NPASS; ++p) {
#define T5_GPOINT(K)                                                                                             \
    {                                                                                                            \
        const int bh = qb5<K>(hl[p]), bxx = qb5<K>(wl[p]);                                                       \
        const float b1 = qb5<K>(w1[p]), b2 = qb5<K>(w2[p]), b3 = qb5<K>(w3[p]), b4 = qb5<K>(w4[p]);              \
        const bool bo = qb5<K>((int)((okmask >> p) & 1u)) != 0;                                                  \
        const bool u0 = bo && bh >= 0, u1 = bo && bh + 1 <= H - 1, l0 = bxx >= 0, l1 = bxx + 1 <= W - 1;         \
        const int h0 = min(max(bh, 0), H - 1), h1 = min(max(bh + 1, 0), H - 1);                                  \
        const int c0 = min(max(bxx, 0), W - 1), c1 = min(max(bxx + 1, 0), W - 1);                                \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h0 * W + c0) * MD));    \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h0 * W + c1) * MD));    \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h1 * W + c0) * MD));    \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(vl + (size_t)((unsigned)(h1 * W + c1) * MD));    \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
            acc2[p][c >> 1][c & 1] += b1 * ((u0 && l0) ? v1[c] : 0.f) + b2 * ((u0 && l1) ? v2[c] : 0.f) +        \
                                      b3 * ((u1 && l0) ? v3[c] : 0.f) + b4 * ((u1 && l1) ? v4[c] : 0.f);         \
        }                                                                                                        \
    }
                        T5_GPOINT(0) T5_GPOINT(1) T5_GPOINT(2) T5_GPOINT(3)
#undef T5_GPOINT
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                continue;
            }
            // the other pass comes from the mirror lane's own pass
            w1[1] = hm5(w1[0]); w2[1] = hm5(w2[0]); w3[1] = hm5(w3[0]); w4[1] = hm5(w4[0]);
            // one ABSOLUTE LDS byte offset per point; a rejected point reads the zero strip (offset 0)
            const int pitchB = ww * 128;
            const int wofs = (T5_ZPX + wbuf * T5_WINPIX) * 128;
            int o[T5_NPASS];
            o[0] = (okmask & 1u) ? wofs + ((hl[0] - y0) * ww + (wl[0] - x0)) * 128 : 0;
            o[1] = hm5(o[0]);
            const char *wrow1 = wbase + pitchB;
#define T5_POINT(K)                                                                                              \
    {                                                                                                            \
        const int off = qb5<K>(o[p]);                                                                            \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(wbase + off);                                    \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(wbase + off + 128);                              \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(wrow1 + off);                                    \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(wrow1 + off + 128);                              \
        const float b1 = qb5<K>(w1[p]), b2 = qb5<K>(w2[p]), b3 = qb5<K>(w3[p]), b4 = qb5<K>(w4[p]);              \
        acc2[p][0] = fma2_5(b4, v4.lo, fma2_5(b3, v3.lo, fma2_5(b2, v2.lo, fma2_5(b1, v1.lo, acc2[p][0]))));      \
        acc2[p][1] = fma2_5(b4, v4.hi, fma2_5(b3, v3.hi, fma2_5(b2, v2.hi, fma2_5(b1, v1.hi, acc2[p][1]))));      \
    }
#pragma unroll
            for (int p = 0; p < T5_NPASS; ++p) {
                T5_POINT(0) T5_POINT(1) T5_POINT(2) T5_POINT(3)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef T5_POINT
        }
#pragma unroll
        for (int p = 0; p < T5_NPASS; ++p) {
            bool ok;
            const unsigned qi = pair_of(cur, p, ok);
            if (ok) {
                float4_t ov = {acc2[p][0].x, acc2[p][0].y, acc2[p][1].x, acc2[p][1].y};
                *reinterpret_cast<float4_t *>(out + (size_t)qi * D + sub * 4) = ov;
            }
        }
        cur = nxt; have = have_next;
        q01 = nq01; qok = nqok;
    }
}

}  // namespace

int msda_tiled5_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                       const float *attw, int B, int S, int M, int L, int Lq, float *out, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled5_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T5_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled5_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)T5_LDS);
        attr_set = true;
    }
    const int grid = (cus / 8) * 8;   // persistent: ONE block per CU (two windows)
    if (msda_tiled_enabled() == 7)
        VLLM_LAUNCH((msda_fwd_tiled5_kernel<8>), dim3(grid), dim3((T5_NC + 8) * 64), T5_LDS, st, value, shapes, lsi, loc, attw, B, S, M,
                    L, Lq, out);
    else
        VLLM_LAUNCH((msda_fwd_tiled5_kernel<4>), dim3(grid), dim3((T5_NC + 4) * 64), T5_LDS, st, value, shapes, lsi, loc, attw, B, S, M,
                    L, Lq, out);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled5_kernel");
    return VLLM_OK;
}

}  // namespace vllm
