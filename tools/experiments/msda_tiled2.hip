// Generation 2 of the LDS-tiled MSDA forward (rounds 1-4: `msda_tiled` 3 and the fallback for tensors whose offsets do not fit 32 bits).
// Out of the library since round 5: such tensors (a value tensor beyond 4 GB) take the gather kernel of msda.hip, whose indices are 64-bit.
// Kept as the measured record (811-890 us at cfg 4, B = 8; DESIGN / NOTES).  Not built.
// MSDA forward, LDS-tiled variant for the encoder self-attention case (queries == pyramid pixels, Lq == S).
//
// Why: the plain gather kernel (msda.hip) is limited by the L2 line-request rate -- 18x more 128-byte lines are
// requested than exist, and the 32 KiB vector L1 cannot hold the overlap between neighbouring queries.  Here a block
// owns an 8x16 tile of queries of ONE head.  For each target level it (1) computes the exact bounding window of all
// corners its 128 x P sampling points touch, (2) stages that window once into LDS with LDS-DMA (lane-linear:
// 8 lanes x 16 B = one 128-byte (pixel, head) row per 8-lane group, 8 pixels per wave instruction), (3) gathers the
// 4 x P corners per query from LDS (256 B/clk/CU instead of the ~30 B/clk/CU the L2 path sustained).  A level
// whose window does not fit the 64 KiB budget simply gathers from global memory for that (block, level): the window
// is computed from the actual sampling locations, so correctness never depends on offsets being small.
// Arithmetic is the SAME instruction sequence per (query, level, point) as msda_fwd_vec_kernel, so the two kernels
// are bit-identical (tested).
//
// Persistent grid: the tile decomposition needs (H, W) of every level, which the reference ABI only provides as a
// DEVICE tensor; rather than a host sync, each block derives the tile table from the device tensor and strides
// through the work items itself (XCD-aware: an XCD walks a contiguous range of (batch, head) slabs).
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

constexpr int MT_THREADS = 256;
constexpr int MT_QPP = MT_THREADS / 8;           // 32 queries per pass (8 lanes x 16 B = D 32 fp32)
constexpr int MT_MAXL = 8;

// Tile configuration: TH x TW queries of one level per block, window budget WIN pixels (x 128 B), BPC blocks per CU.
//   <8,16,560,2>: big tiles, fewer halo re-reads, 2 blocks (8 waves) per CU
//   <8, 8,288,4>: small tiles -> small windows and half the per-lane state -> 4 blocks (16 waves) per CU
template <int TH_, int TW_, int WIN_, int BPC_>
struct MTCfg {
    static constexpr int TH = TH_, TW = TW_, NQ = TH_ * TW_, NPASS = NQ / MT_QPP, WIN_MAX = WIN_, ZP = WIN_ + 8, BPC = BPC_;
    static constexpr size_t LDS_WIN = (size_t)(ZP + 1) * 128, LDS_LOC = (size_t)NQ * 4 * 8, LDS_AW = (size_t)NQ * 4 * 4;
    static constexpr size_t LDS = LDS_WIN + LDS_LOC + LDS_AW;
};

template <int K>
__device__ __forceinline__ float quad_bcast(float x)   // value of lane K of this lane's quad
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ int quad_bcast(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, false);
}

// Global-memory fallback for one (block, level) whose window exceeds the LDS budget: the gather kernel's code,
// with loc / weights read from the LDS copy.
template <int PT, typename Cfg>
__device__ __forceinline__ void gather_level_global(float (&acc)[Cfg::NPASS][4], const bool (&qok)[Cfg::NPASS],
                                                    const float2_t *s_loc, const float *s_aw, int slot0, int H, int W,
                                                    const float *__restrict__ vl, long MD)
{
#pragma unroll
    for (int p = 0; p < Cfg::NPASS; ++p) {
        const int slot = p * MT_QPP + slot0;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const float2_t xy = s_loc[slot * PT + k];
            const float aw = s_aw[slot * PT + k];
            const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
            const int hl = sp.h_low, wl = sp.w_low;
            const float lh = sp.h_im - (float)hl, lw = sp.w_im - (float)wl;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool pok = sp.ok && qok[p] && H > 0 && W > 0;
            const float w1 = pok ? hh * hw : 0.f, w2 = pok ? hh * lw : 0.f, w3 = pok ? lh * hw : 0.f, w4 = pok ? lh * lw : 0.f;
            const bool k1 = pok && hl >= 0 && wl >= 0;
            const bool k2 = pok && hl >= 0 && wl + 1 <= W - 1;
            const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0;
            const bool k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;
            const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
            const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);
            const float4_t v1 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x0) * MD);
            const float4_t v2 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x1) * MD);
            const float4_t v3 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x0) * MD);
            const float4_t v4 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x1) * MD);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f;
                const float a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;
                const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;
                acc[p][c] += val * aw;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int PT, typename Cfg>
__global__ __launch_bounds__(MT_THREADS, Cfg::BPC) void msda_fwd_tiled_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out)
{
    static_assert(PT == 4, "one sampling point per lane of a quad");
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *win = reinterpret_cast<float *>(smem);                               // [(Cfg::ZP + 1) pixels][32]
    float2_t *s_loc = reinterpret_cast<float2_t *>(smem + Cfg::LDS_WIN);            // [128 queries][4 points]
    float *s_aw = reinterpret_cast<float *>(smem + Cfg::LDS_WIN + Cfg::LDS_LOC);      // [128 queries][4 points]
    __shared__ int s_H[MT_MAXL], s_W[MT_MAXL], s_q0[MT_MAXL], s_tc[MT_MAXL + 1];
    __shared__ long s_v0[MT_MAXL];
    __shared__ int s_red[4][4];
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = tid & 7;                     // 16-byte channel chunk of this lane
    const int kpt = tid & 3;                     // the sampling point this lane evaluates for its quad
    const int slot0 = tid >> 3;                  // query slot inside a pass
    const long MD = (long)M * D;

    // ---- tile table from the device-side shapes; zero pixel ----
    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + Cfg::TH - 1) / Cfg::TH) * ((W + Cfg::TW - 1) / Cfg::TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    if (tid < 32) win[Cfg::ZP * 32 + tid] = 0.f;
    __syncthreads();
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + Cfg::TW - 1) / Cfg::TW;
    const long n_items = (long)B * M * n_tiles;

    const int xcd = blockIdx.x & 7;
    const long ipx = (n_items + 7) >> 3;
    const int blocks_per_xcd = gridDim.x >> 3;

    for (long j = blockIdx.x >> 3; j < ipx; j += blocks_per_xcd) {
        const long item = (long)xcd * ipx + j;
        if (item >= n_items) break;
        const int t = (int)(item % n_tiles);
        const long bm = item / n_tiles;
        const int m = (int)(bm % M);
        const long b = bm / M;
        int qH, qW, q0, ty, tx;
        if (geo) {
            int lq = 0;
            while (lq + 1 < L && s_tc[lq + 1] <= t) ++lq;
            qH = s_H[lq]; qW = s_W[lq]; q0 = s_q0[lq];
            const int txn = (qW + Cfg::TW - 1) / Cfg::TW, tl = t - s_tc[lq];
            ty = tl / txn; tx = tl - ty * txn;
        } else {
            qH = 1; qW = Lq; q0 = 0; ty = 0; tx = t;
        }
        // (b, q, m) pair index of tile slot s (clamped to a live query) and whether the slot is live
        auto pair_of = [&](int slot, bool &ok) -> long {
            const int y = ty * Cfg::TH + slot / Cfg::TW, x = tx * Cfg::TW + slot % Cfg::TW;
            ok = y < qH && x < qW;
            const long q = q0 + (long)(ok ? y : 0) * qW + (ok ? x : 0);
            return (b * Lq + q) * M + m;
        };

        long qidx[Cfg::NPASS];
        bool qok[Cfg::NPASS];
        float acc[Cfg::NPASS][4];
#pragma unroll
        for (int p = 0; p < Cfg::NPASS; ++p) {
            qidx[p] = pair_of(p * MT_QPP + slot0, qok[p]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[p][c] = 0.f;
        }
        // cooperative loc / weight loads: thread i -> query slot i>>1, points 2*(i&1).. ; weights: threads < 128
        bool lq_ok, aq_ok;
        const long lq_pair = pair_of(tid >> 1, lq_ok);
        const long aq_pair = pair_of(tid & (Cfg::NQ - 1), aq_ok);

        // loc / weights of level l+1 are fetched into registers while level l is processed (the global latency would
        // otherwise be exposed twice per level: once here, once for the window)
        const bool lthr = tid < Cfg::NQ * 2;
        float4_t nloc = {0.f, 0.f, 0.f, 0.f};
        if (lthr) nloc = *reinterpret_cast<const float4_t *>(loc + (lq_pair * L + 0) * (PT * 2) + (tid & 1) * 4);
        float4_t naw = {0.f, 0.f, 0.f, 0.f};
        if (tid < Cfg::NQ) naw = *reinterpret_cast<const float4_t *>(attw + (aq_pair * L + 0) * PT);

        for (int l = 0; l < L; ++l) {
            const int H = s_H[l], W = s_W[l];
            const float *vl = value + (b * (long)S + s_v0[l]) * MD + (long)m * D + sub * 4;

            __syncthreads();   // previous level / item: every read of s_loc, s_aw, win, s_red is finished
            if (lthr) reinterpret_cast<float4_t *>(s_loc)[tid] = nloc;
            if (tid < Cfg::NQ) reinterpret_cast<float4_t *>(s_aw)[tid] = naw;
            if (l + 1 < L) {
                if (lthr) nloc = *reinterpret_cast<const float4_t *>(loc + (lq_pair * L + l + 1) * (PT * 2) + (tid & 1) * 4);
                if (tid < Cfg::NQ) naw = *reinterpret_cast<const float4_t *>(attw + (aq_pair * L + l + 1) * PT);
            }
            __syncthreads();

            // ---- A: this lane's point (kpt) of each of its 4 queries; exact bounding window of all corners ----
            float him[Cfg::NPASS], wim[Cfg::NPASS], awp[Cfg::NPASS];
            int hlo[Cfg::NPASS], wlo[Cfg::NPASS];
            bool okp[Cfg::NPASS];
            int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
#pragma unroll
            for (int p = 0; p < Cfg::NPASS; ++p) {
                const int slot = p * MT_QPP + slot0;
                const float2_t xy = s_loc[slot * PT + kpt];
                awp[p] = s_aw[slot * PT + kpt];
                const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
                him[p] = sp.h_im; wim[p] = sp.w_im; hlo[p] = sp.h_low; wlo[p] = sp.w_low;
                okp[p] = sp.ok && qok[p] && H > 0 && W > 0;   // (empty level: no corner inside, adds nothing)
                if (okp[p]) {
                    const int h0 = min(max(sp.h_low, 0), H - 1), h1 = min(max(sp.h_low + 1, 0), H - 1);
                    const int x0 = min(max(sp.w_low, 0), W - 1), x1 = min(max(sp.w_low + 1, 0), W - 1);
                    ymin = min(ymin, h0); ymax = max(ymax, h1); xmin = min(xmin, x0); xmax = max(xmax, x1);
                }
            }
            int r0 = ymin, r1 = -ymax, r2 = xmin, r3 = -xmax;   // four min-reductions
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                r0 = min(r0, __shfl_xor(r0, o)); r1 = min(r1, __shfl_xor(r1, o));
                r2 = min(r2, __shfl_xor(r2, o)); r3 = min(r3, __shfl_xor(r3, o));
            }
            if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; }
            __syncthreads();
            const int y0 = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
            const int y1 = -min(min(s_red[0][1], s_red[1][1]), min(s_red[2][1], s_red[3][1]));
            const int x0w = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
            const int x1w = -min(min(s_red[0][3], s_red[1][3]), min(s_red[2][3], s_red[3][3]));
            if (y1 < 0) continue;                        // no accepted point at this level (block-uniform)
            const int wh = y1 - y0 + 1, ww = x1w - x0w + 1;
            const int npix = wh * ww;
            if (npix > Cfg::WIN_MAX) {                     // block-uniform: window does not fit -> gather from global
                gather_level_global<PT, Cfg>(acc, qok, s_loc, s_aw, slot0, H, W, vl, MD);
                continue;
            }

            // ---- B: stage the window (LDS-DMA, 8 pixels of 128 B per wave instruction) ----
            const unsigned ww_magic = (1u << 20) / (unsigned)ww + 1u;   // one scalar division per (block, level)
            for (int i0 = wave * 8; i0 < npix; i0 += 32) {
                int pix = i0 + (lane >> 3);
                pix = pix < npix ? pix : npix - 1;
                const int wy = (int)(((unsigned)pix * ww_magic) >> 20), wx = pix - wy * ww;   // pix / ww, exact (pix*ww < 2^20)
                const float *g = vl + ((long)(y0 + wy) * W + (x0w + wx)) * MD;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
            }
            // the owner lane turns its point into 4 LDS byte offsets (+ this lane's channel chunk) and 4 weights while
            // the DMA is in flight; a corner that must not contribute points at the all-zero pixel
            int o1[Cfg::NPASS], o2[Cfg::NPASS], o3[Cfg::NPASS], o4[Cfg::NPASS];
            float w1[Cfg::NPASS], w2[Cfg::NPASS], w3[Cfg::NPASS], w4[Cfg::NPASS];
#pragma unroll
            for (int p = 0; p < Cfg::NPASS; ++p) {
                const int hl = hlo[p], wl = wlo[p];
                const float lh = him[p] - (float)hl, lw = wim[p] - (float)wl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool pok = okp[p];
                w1[p] = pok ? hh * hw : 0.f; w2[p] = pok ? hh * lw : 0.f;
                w3[p] = pok ? lh * hw : 0.f; w4[p] = pok ? lh * lw : 0.f;
                const bool k1 = pok && hl >= 0 && wl >= 0;
                const bool k2 = pok && hl >= 0 && wl + 1 <= W - 1;
                const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0;
                const bool k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;
                const int ry0 = hl - y0, ry1 = hl + 1 - y0, rx0 = wl - x0w, rx1 = wl + 1 - x0w;
                o1[p] = (k1 ? ry0 * ww + rx0 : Cfg::ZP) * 128;
                o2[p] = (k2 ? ry0 * ww + rx1 : Cfg::ZP) * 128;
                o3[p] = (k3 ? ry1 * ww + rx0 : Cfg::ZP) * 128;
                o4[p] = (k4 ? ry1 * ww + rx1 : Cfg::ZP) * 128;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            // ---- C: gather from LDS; the quad's lane K broadcasts point K's offsets / weights (DPP) ----
            const char *wbase = reinterpret_cast<const char *>(win) + sub * 16;
#define MT_POINT(K)                                                                                              \
    {                                                                                                            \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(wbase + quad_bcast<K>(o1[p]));                   \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(wbase + quad_bcast<K>(o2[p]));                   \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(wbase + quad_bcast<K>(o3[p]));                   \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(wbase + quad_bcast<K>(o4[p]));                   \
        const float b1 = quad_bcast<K>(w1[p]), b2 = quad_bcast<K>(w2[p]), b3 = quad_bcast<K>(w3[p]),             \
                    b4 = quad_bcast<K>(w4[p]), ba = quad_bcast<K>(awp[p]);                                       \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
            const float val = b1 * v1[c] + b2 * v2[c] + b3 * v3[c] + b4 * v4[c];                                 \
            acc[p][c] += val * ba;                                                                               \
        }                                                                                                        \
    }
#pragma unroll
            for (int p = 0; p < Cfg::NPASS; ++p) {
                MT_POINT(0) MT_POINT(1) MT_POINT(2) MT_POINT(3)
            }
#undef MT_POINT
        }
#pragma unroll
        for (int p = 0; p < Cfg::NPASS; ++p)
            if (qok[p]) {
                float4_t o = {acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                *reinterpret_cast<float4_t *>(out + qidx[p] * D + sub * 4) = o;
            }
    }
}

}  // namespace vllm
