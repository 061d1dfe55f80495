// MSDA forward, LDS-tiled AND software-pipelined (encoder self-attention shape).  Same tiling idea and the same
// per-point arithmetic as msda_tiled.hip, but the three latencies that kernel exposes per (tile, level) step --
// sampling locations, bounding-window reduction, window staging -- are taken off the critical path:
//
//   step s = (item k, level l) of a persistent block (8 waves, one block per CU, 152 KiB of LDS)
//   iteration s:   [sync] loc(s+1): registers -> LDS ; global loads of loc(s+2) -> registers
//                  [sync] prepare(s+1): each quad lane evaluates ONE sampling point of its query, bbox partials
//                  [sync] window(s+1) -> LDS-DMA into win[(s+1)&1]   (asynchronous)
//                         offsets / weights of step s+1 -> registers (descriptor D_nxt)
//                         gather(s) from win[s&1] with D_cur  (LDS + VALU only: overlaps the DMA just issued)
//                         s_waitcnt vmcnt(0)
// Correctness argument for the two window buffers: win[(s+1)&1] was last read by gather(s-1), which every wave
// finished before the first barrier of iteration s; the DMA of step s was waited for (vmcnt(0)) by every issuing wave at
// the end of iteration s-1 and published by that same barrier.
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

constexpr int MP_TH = 8, MP_TW = 16;             // query tile
constexpr int MP_THREADS = 512;
constexpr int MP_WAVES = MP_THREADS / 64;
constexpr int MP_NQ = MP_TH * MP_TW;             // 128 queries
constexpr int MP_QPP = MP_THREADS / 8;           // 64 query slots per pass
constexpr int MP_NPASS = MP_NQ / MP_QPP;         // 2
constexpr int MP_WIN_MAX = 560;
constexpr int MP_WIN_PIX = MP_WIN_MAX + 8;       // + LDS-DMA slack
constexpr int MP_MAXL = 8;
constexpr size_t MP_LDS_WIN = (size_t)MP_WIN_PIX * 128;              // one window buffer
constexpr size_t MP_OFF_ZERO = 2 * MP_LDS_WIN;                       // all-zero pixel behind the two windows
constexpr size_t MP_OFF_LOC = MP_OFF_ZERO + 128;
constexpr size_t MP_OFF_AW = MP_OFF_LOC + MP_NQ * 4 * 8;
constexpr size_t MP_LDS = MP_OFF_AW + MP_NQ * 4 * 4;

template <int K>
__device__ __forceinline__ float qb(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ int qb(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, false);
}

struct StepGeo {          // block-uniform description of one step
    int valid;            // step exists
    int l;                // target level
    int m;                // head
    long b;               // batch index
    int qH, qW, q0, ty, tx;
};

struct StepDesc {         // per-lane descriptor of one step (built by prepare, consumed by gather)
    int o1[MP_NPASS], o2[MP_NPASS], o3[MP_NPASS], o4[MP_NPASS];   // LDS byte offsets of the 4 corners (or the zero pixel)
    float w1[MP_NPASS], w2[MP_NPASS], w3[MP_NPASS], w4[MP_NPASS];
    float aw[MP_NPASS];
    float him[MP_NPASS], wim[MP_NPASS];                           // raw sample (global fallback only)
    int hlo[MP_NPASS], wlo[MP_NPASS];
    bool ok[MP_NPASS];
    int mode;             // 0: nothing to do, 1: window in LDS, 2: gather from global (window too large)
};

__global__ __launch_bounds__(MP_THREADS, 1) void msda_fwd_pipe_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out)
{
    constexpr int D = 32, PT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2_t *s_loc = reinterpret_cast<float2_t *>(smem + MP_OFF_LOC);
    float *s_aw = reinterpret_cast<float *>(smem + MP_OFF_AW);
    __shared__ int s_H[MP_MAXL], s_W[MP_MAXL], s_q0[MP_MAXL], s_tc[MP_MAXL + 1];
    __shared__ long s_v0[MP_MAXL];
    __shared__ int s_red[MP_WAVES][4];
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = tid & 7, kpt = tid & 3, slot0 = tid >> 3;
    const long MD = (long)M * D;

    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + MP_TH - 1) / MP_TH) * ((W + MP_TW - 1) / MP_TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    if (tid < 32) reinterpret_cast<float *>(smem + MP_OFF_ZERO)[tid] = 0.f;
    __syncthreads();
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + MP_TW - 1) / MP_TW;
    const long n_items = (long)B * M * n_tiles;

    // XCD-aware persistent walk: XCD x owns the contiguous item range [x*ipx, (x+1)*ipx)
    const int xcd = blockIdx.x & 7;
    const long ipx = (n_items + 7) >> 3;
    const int bpx = gridDim.x >> 3;
    const long first = blockIdx.x >> 3;
    long my_items = 0;
    {
        const long lim = min(ipx, n_items - (long)xcd * ipx);   // items of this XCD
        if (first < lim) my_items = (lim - first + bpx - 1) / bpx;
    }
    const long T = my_items * L;   // steps of this block

    // step -> geometry.  32-bit arithmetic only (64-bit integer division costs hundreds of VALU instructions and this is
    // evaluated by every lane once per step).
    const int T32 = (int)T, item0 = (int)((long)xcd * ipx + first);
    auto decode = [&](long s) -> StepGeo {
        StepGeo g;
        g.valid = s < T;
        const int sc = g.valid ? (int)s : 0;
        const int k = sc / L;
        g.l = sc - k * L;
        const unsigned item = (unsigned)(item0 + k * bpx);
        const unsigned bm = item / (unsigned)n_tiles;
        const int t = (int)(item - bm * (unsigned)n_tiles);
        const unsigned bb = bm / (unsigned)M;
        g.m = (int)(bm - bb * (unsigned)M);
        g.b = (long)bb;
        if (geo) {
            int lq = 0;
            while (lq + 1 < L && s_tc[lq + 1] <= t) ++lq;
            g.qH = s_H[lq]; g.qW = s_W[lq]; g.q0 = s_q0[lq];
            const int txn = (g.qW + MP_TW - 1) / MP_TW, tl = t - s_tc[lq];
            g.ty = tl / txn; g.tx = tl - g.ty * txn;
        } else {
            g.qH = 1; g.qW = Lq; g.q0 = 0; g.ty = 0; g.tx = t;
        }
        return g;
    };
    (void)T32;
    auto pair_of = [&](const StepGeo &g, int slot, bool &ok) -> long {
        const int y = g.ty * MP_TH + slot / MP_TW, x = g.tx * MP_TW + slot % MP_TW;
        ok = y < g.qH && x < g.qW;
        const long q = g.q0 + (long)(ok ? y : 0) * g.qW + (ok ? x : 0);
        return (g.b * Lq + q) * M + g.m;
    };
    // loc / weights of one step -> registers (thread i: query slot i>>2, point i&3; weights: query i>>2 ... as float)
    // 512 threads x float2 = the 128 x 4 (x, y) pairs; threads < 128 also fetch the 4 weights of query `tid`.
    auto fetch_loc = [&](const StepGeo &g, float2_t &xy, float4_t &aw4) {
        bool ok;
        const long pr = pair_of(g, tid >> 2, ok);
        xy = *reinterpret_cast<const float2_t *>(loc + ((pr * L + g.l) * PT + (tid & 3)) * 2);
        if (tid < MP_NQ) {
            const long pa = pair_of(g, tid, ok);
            aw4 = *reinterpret_cast<const float4_t *>(attw + (pa * L + g.l) * PT);
        }
    };

    float acc[MP_NPASS][4];
    StepDesc dc, dn;       // descriptor being consumed / being built
    dc.mode = 0;
    StepGeo gc, gn, gp;    // geometry of step s (gather), s+1 (prepare), s+2 (loc prefetch)
    gc.valid = 0;
    gn = decode(0);
    gp = decode(1);
    float2_t pxy = {0.f, 0.f};
    float4_t paw = {0.f, 0.f, 0.f, 0.f};
    if (gn.valid) fetch_loc(gn, pxy, paw);     // loc of step 0

    for (long s = -1; s < T; ++s) {
        // ================= top of iteration s =================
        __syncthreads();   // S1: gather(s-1) and prepare(s) are finished in every wave
        if (gn.valid) {
            s_loc[tid] = pxy;                                  // [query slot][point] == tid
            if (tid < MP_NQ) reinterpret_cast<float4_t *>(s_aw)[tid] = paw;
        }
        if (gp.valid) fetch_loc(gp, pxy, paw);                 // loc of step s+2 travels during this iteration
        __syncthreads();   // S2: loc(s+1) visible

        // ---------------- prepare A(s+1): own point of each of the 2 queries, bbox partials ----------------
        int H1 = 1, W1 = 1;
        bool nq_ok[MP_NPASS] = {false, false};
        int r0 = 0x7fffffff, r1 = 1, r2 = 0x7fffffff, r3 = 1;   // min(y), -max(y), min(x), -max(x)   (empty: max = -1)
        if (gn.valid) {
            H1 = s_H[gn.l]; W1 = s_W[gn.l];
#pragma unroll
            for (int p = 0; p < MP_NPASS; ++p) {
                const int slot = p * MP_QPP + slot0;
                bool qk;
                (void)pair_of(gn, slot, qk);
                nq_ok[p] = qk;
                const float2_t xy = s_loc[slot * PT + kpt];
                dn.aw[p] = s_aw[slot * PT + kpt];
                const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H1, W1);
                dn.him[p] = sp.h_im; dn.wim[p] = sp.w_im; dn.hlo[p] = sp.h_low; dn.wlo[p] = sp.w_low;
                dn.ok[p] = sp.ok && qk;
                if (dn.ok[p]) {
                    const int h0 = min(max(sp.h_low, 0), H1 - 1), h1 = min(max(sp.h_low + 1, 0), H1 - 1);
                    const int x0 = min(max(sp.w_low, 0), W1 - 1), x1 = min(max(sp.w_low + 1, 0), W1 - 1);
                    r0 = min(r0, h0); r1 = min(r1, -h1); r2 = min(r2, x0); r3 = min(r3, -x1);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            r0 = min(r0, __shfl_xor(r0, o)); r1 = min(r1, __shfl_xor(r1, o));
            r2 = min(r2, __shfl_xor(r2, o)); r3 = min(r3, __shfl_xor(r3, o));
        }
        if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; }
        __syncthreads();   // S3: bbox partials visible

        // ---------------- prepare B(s+1): window geometry, LDS-DMA, descriptor ----------------
        dn.mode = 0;
        if (gn.valid) {
            int y0 = s_red[0][0], ny1 = s_red[0][1], x0w = s_red[0][2], nx1 = s_red[0][3];
#pragma unroll
            for (int w = 1; w < MP_WAVES; ++w) {
                y0 = min(y0, s_red[w][0]); ny1 = min(ny1, s_red[w][1]);
                x0w = min(x0w, s_red[w][2]); nx1 = min(nx1, s_red[w][3]);
            }
            const int y1 = -ny1, x1w = -nx1;
            if (y1 >= 0) {                                     // some accepted point
                const int wh = y1 - y0 + 1, ww = x1w - x0w + 1, npix = wh * ww;
                if (npix <= MP_WIN_MAX) {
                    dn.mode = 1;
                    const float *vl = value + (gn.b * (long)S + s_v0[gn.l]) * MD + (long)gn.m * D + sub * 4;
                    float *wn = reinterpret_cast<float *>(smem + (size_t)((s + 1) & 1) * MP_LDS_WIN);
                    const unsigned ww_magic = (1u << 20) / (unsigned)ww + 1u;
                    for (int i0 = wave * 8; i0 < npix; i0 += MP_WAVES * 8) {
                        int pix = i0 + (lane >> 3);
                        pix = pix < npix ? pix : npix - 1;
                        const int wy = (int)(((unsigned)pix * ww_magic) >> 20), wx = pix - wy * ww;   // pix / ww, exact

                        const float *g = vl + ((long)(y0 + wy) * W1 + (x0w + wx)) * MD;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                         (__attribute__((address_space(3))) void *)(wn + i0 * 32), 16, 0, 0);
                    }
                    const int wbase = (int)(((s + 1) & 1) * MP_LDS_WIN);     // this lane's channel chunk (sub*16) is added by
                    const int zoff = (int)MP_OFF_ZERO;                        // the READING lane, after the quad broadcast
#pragma unroll
                    for (int p = 0; p < MP_NPASS; ++p) {
                        const int hl = dn.hlo[p], wl = dn.wlo[p];
                        const float lh = dn.him[p] - (float)hl, lw = dn.wim[p] - (float)wl;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const bool pok = dn.ok[p];
                        dn.w1[p] = pok ? hh * hw : 0.f; dn.w2[p] = pok ? hh * lw : 0.f;
                        dn.w3[p] = pok ? lh * hw : 0.f; dn.w4[p] = pok ? lh * lw : 0.f;
                        const bool k1 = pok && hl >= 0 && wl >= 0;
                        const bool k2 = pok && hl >= 0 && wl + 1 <= W1 - 1;
                        const bool k3 = pok && hl + 1 <= H1 - 1 && wl >= 0;
                        const bool k4 = pok && hl + 1 <= H1 - 1 && wl + 1 <= W1 - 1;
                        const int ry0 = hl - y0, ry1 = hl + 1 - y0, rx0 = wl - x0w, rx1 = wl + 1 - x0w;
                        dn.o1[p] = k1 ? wbase + (ry0 * ww + rx0) * 128 : zoff;
                        dn.o2[p] = k2 ? wbase + (ry0 * ww + rx1) * 128 : zoff;
                        dn.o3[p] = k3 ? wbase + (ry1 * ww + rx0) * 128 : zoff;
                        dn.o4[p] = k4 ? wbase + (ry1 * ww + rx1) * 128 : zoff;
                    }
                } else {
                    dn.mode = 2;
                }
            }
        }

        // ---------------- gather(s) ----------------
        if (gc.valid) {
            if (gc.l == 0) {
#pragma unroll
                for (int p = 0; p < MP_NPASS; ++p)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[p][c] = 0.f;
            }
            if (dc.mode == 1) {
                const char *lbase = smem + sub * 16;
#define MP_POINT(K)                                                                                         \
    {                                                                                                       \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(lbase + qb<K>(dc.o1[p]));                   \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(lbase + qb<K>(dc.o2[p]));                   \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(lbase + qb<K>(dc.o3[p]));                   \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(lbase + qb<K>(dc.o4[p]));                   \
        const float b1 = qb<K>(dc.w1[p]), b2 = qb<K>(dc.w2[p]), b3 = qb<K>(dc.w3[p]), b4 = qb<K>(dc.w4[p]); \
        const float ba = qb<K>(dc.aw[p]);                                                                   \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                     \
            const float val = b1 * v1[c] + b2 * v2[c] + b3 * v3[c] + b4 * v4[c];                            \
            acc[p][c] += val * ba;                                                                          \
        }                                                                                                   \
    }
#pragma unroll
                for (int p = 0; p < MP_NPASS; ++p) { MP_POINT(0) MP_POINT(1) MP_POINT(2) MP_POINT(3) }
#undef MP_POINT
            } else if (dc.mode == 2) {
                // window too large for LDS: the gather kernel's path, sample broadcast from the owner lane of the quad
                const int H = s_H[gc.l], W = s_W[gc.l];
                const float *vl = value + (gc.b * (long)S + s_v0[gc.l]) * MD + (long)gc.m * D + sub * 4;
#define MP_GPOINT(K)                                                                                        \
    {                                                                                                       \
        const int hl = qb<K>(dc.hlo[p]), wl = qb<K>(dc.wlo[p]);                                             \
        const float hi = qb<K>(dc.him[p]), wi = qb<K>(dc.wim[p]), ba = qb<K>(dc.aw[p]);                     \
        const bool pok = qb<K>((int)dc.ok[p]) != 0;                                                         \
        const float lh = hi - (float)hl, lw = wi - (float)wl, hh = 1.f - lh, hw = 1.f - lw;                 \
        const float b1 = pok ? hh * hw : 0.f, b2 = pok ? hh * lw : 0.f, b3 = pok ? lh * hw : 0.f,           \
                    b4 = pok ? lh * lw : 0.f;                                                               \
        const bool k1 = pok && hl >= 0 && wl >= 0, k2 = pok && hl >= 0 && wl + 1 <= W - 1;                  \
        const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0, k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;  \
        const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);                             \
        const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);                             \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x0) * MD);             \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x1) * MD);             \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x0) * MD);             \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x1) * MD);             \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                     \
            const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f, a3 = k3 ? v3[c] : 0.f,                \
                        a4 = k4 ? v4[c] : 0.f;                                                              \
            const float val = b1 * a1 + b2 * a2 + b3 * a3 + b4 * a4;                                        \
            acc[p][c] += val * ba;                                                                          \
        }                                                                                                   \
    }
#pragma unroll
                for (int p = 0; p < MP_NPASS; ++p) { MP_GPOINT(0) MP_GPOINT(1) MP_GPOINT(2) MP_GPOINT(3) }
#undef MP_GPOINT
            }
            if (gc.l == L - 1) {
#pragma unroll
                for (int p = 0; p < MP_NPASS; ++p) {
                    bool qk;
                    const long pr = pair_of(gc, p * MP_QPP + slot0, qk);
                    if (qk) {
                        float4_t o = {acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                        *reinterpret_cast<float4_t *>(out + pr * D + sub * 4) = o;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of step s+1 (and loc of s+2) has landed
        // rotate; the next level of the same item only bumps l, a new item is decoded from scratch
        dc = dn;
        gc = gn; gn = gp;
        if (gp.valid && gp.l + 1 < L && s + 3 < T) gp.l += 1;
        else gp = decode(s + 3);
    }
}

int msda_pipe_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                     const float *attw, int B, int S, int M, int L, int Lq, float *out, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const size_t lds = MP_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = (cus / 8) * 8;   // persistent: one 8-wave block per CU
    VLLM_LAUNCH(msda_fwd_pipe_kernel, dim3(grid), dim3(MP_THREADS), lds, st, value, shapes, lsi, loc, attw, B, S, M, L, Lq, out);
    VLLM_CHECK_LAUNCH("msda_fwd_pipe_kernel");
    return VLLM_OK;
}

}  // namespace vllm
