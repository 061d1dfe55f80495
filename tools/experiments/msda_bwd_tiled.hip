// MSDA backward, LDS-tiled variant for the encoder self-attention case (queries == pyramid pixels, Lq == S, D = 32, P = 4).
//
// Follows ms_deform_im2col_cuda.cuh:87-161 (col2im bilinear: grad_value scatter, grad_sampling_loc, grad_attn_weight) and
// the shared-memory reductions of :301-360; the plain kernel (msda.hip, msda_bwd_vec_kernel) issues one hardware fp32
// atomic per (point, corner, channel) straight to global memory -- 4.9 G scattered 4-byte atomics at the cfg-4 encoder
// shape, which is what its 57 ms are.  Here a block owns an 8x16 tile of queries of ONE head (the forward kernel's
// decomposition, msda_tiled.hip).  For each target level it
//   (1) computes the exact bounding window of all corners its 128 x 4 sampling points touch,
//   (2) accumulates grad_value for that window in LDS (ds_add_f32, window <= 560 pixels x 32 channels fp32),
//   (3) flushes the window once: one atomic per (window pixel, channel), 128 contiguous bytes per pixel.
// Neighbouring queries hit the same pixels, so the window holds ~4x fewer elements than there are (point, corner)
// contributions, and the global atomics that remain are whole-line instead of 16-byte-strided.  grad_sampling_loc and
// grad_attn_weight have exactly one owner (plain stores), the value corners they need are read from global memory / L2 as
// in the plain kernel.  A level whose window does not fit falls back to direct global atomics for that (block, level):
// correctness never depends on offsets being small.  Summation order differs from the plain kernel (both are atomic
// scatters); tests compare against the oracle with the same tolerance.
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

// Timing-only ablation builds (tools/msda_bwd_ablate.sh): -DBT_ABL=<mask>.  1: no flush atomics, 2: no direct (cold-window) atomics,
// 4: no LDS adds, 8: no value corner reads (zeros), 16: no grad_loc / grad_attw stores, 32: no window clear, 64: no flush loop.
#ifndef BT_ABL
#define BT_ABL 0
#endif

#ifdef BT_PROF
#define BT_TICK(slot) { const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime(); pacc[slot] += now__ - tprev; tprev = now__; }
#else
#define BT_TICK(slot)
#endif

namespace vllm {
namespace {

#ifdef BT_PROF
__device__ unsigned long long g_bt_prof[16];   // phase clock of the diagnostics build (tools/msda_bwd_ablate.sh): ticks of wave 0 of every block
#endif

constexpr int BT_THREADS = 256;
constexpr int BT_QPP = BT_THREADS / 8;   // 32 queries per pass (8 lanes x 4 channels = D 32)
constexpr int BT_MAXL = 8;
constexpr int BT_TH = 8, BT_TW = 16, BT_NQ = BT_TH * BT_TW, BT_NPASS = BT_NQ / BT_QPP, BT_WIN = 360;   // 360-pixel windows: 3 blocks per CU (see msda_tiled4_launch)
constexpr size_t BT_LDS_WIN = (size_t)BT_WIN * 128, BT_LDS_LOC = (size_t)BT_NQ * 4 * 8, BT_LDS_AW = (size_t)BT_NQ * 4 * 4;
constexpr size_t BT_LDS = BT_LDS_WIN + BT_LDS_LOC + BT_LDS_AW;

template <int K> __device__ __forceinline__ float qbc(float x)   // value of lane K of this lane's quad
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ int qbc(int x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, false); }

__global__ __launch_bounds__(BT_THREADS, 3) void msda_bwd_tiled_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out, int B, int S, int M,
    int L, int Lq, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attw)
{
    constexpr int D = 32, PT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *gwin = reinterpret_cast<float *>(smem);                                 // [<= 560 pixels][32] grad_value window
    float2_t *s_loc = reinterpret_cast<float2_t *>(smem + BT_LDS_WIN);             // [128 queries][4 points]
    float *s_aw = reinterpret_cast<float *>(smem + BT_LDS_WIN + BT_LDS_LOC);       // [128 queries][4 points]
    __shared__ int s_H[BT_MAXL], s_W[BT_MAXL], s_q0[BT_MAXL], s_tc[BT_MAXL + 1];
    __shared__ long s_v0[BT_MAXL];
    __shared__ int s_red[4][4];
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = tid & 7;      // 4-channel chunk of this lane
    const int kpt = tid & 3;      // the sampling point this lane evaluates for its quad
    const int slot0 = tid >> 3;   // query slot inside a pass
    const long MD = (long)M * D;

    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + BT_TH - 1) / BT_TH) * ((W + BT_TW - 1) / BT_TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    __syncthreads();
#ifdef BT_PROF
    unsigned pacc[16] = {};
    unsigned tprev = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + BT_TW - 1) / BT_TW;
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
            const int txn = (qW + BT_TW - 1) / BT_TW, tl = t - s_tc[lq];
            ty = tl / txn; tx = tl - ty * txn;
        } else {
            qH = 1; qW = Lq; q0 = 0; ty = 0; tx = t;
        }
        auto pair_of = [&](int slot, bool &ok) -> long {
            const int y = ty * BT_TH + slot / BT_TW, x = tx * BT_TW + slot % BT_TW;
            ok = y < qH && x < qW;
            const long q = q0 + (long)(ok ? y : 0) * qW + (ok ? x : 0);
            return (b * Lq + q) * M + m;
        };

        long qidx[BT_NPASS];
        bool qok[BT_NPASS];
        float4_t go[BT_NPASS];   // this lane's 4 channels of grad_output of its query slots
#pragma unroll
        for (int p = 0; p < BT_NPASS; ++p) {
            qidx[p] = pair_of(p * BT_QPP + slot0, qok[p]);
            go[p] = *reinterpret_cast<const float4_t *>(grad_out + qidx[p] * D + sub * 4);
            if (!qok[p]) go[p] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        bool lq_ok, aq_ok;
        const long lq_pair = pair_of(tid >> 1, lq_ok);
        const long aq_pair = pair_of(tid & (BT_NQ - 1), aq_ok);
        const bool lthr = tid < BT_NQ * 2;
        float4_t nloc = {0.f, 0.f, 0.f, 0.f};
        if (lthr) nloc = *reinterpret_cast<const float4_t *>(loc + (lq_pair * L + 0) * (PT * 2) + (tid & 1) * 4);
        float4_t naw = {0.f, 0.f, 0.f, 0.f};
        if (tid < BT_NQ) naw = *reinterpret_cast<const float4_t *>(attw + (aq_pair * L + 0) * PT);

        BT_TICK(0)   // item set-up: decode, grad_output / first level's locations requested
        for (int l = 0; l < L; ++l) {
            const int H = s_H[l], W = s_W[l];
            const long lbase = (b * (long)S + s_v0[l]) * MD + (long)m * D;   // (batch, level, head) origin, channel 0
            const float *vl = value + lbase + sub * 4;

            __syncthreads();   // previous level / item: every read of s_loc, s_aw, gwin, s_red is finished
            if (lthr) reinterpret_cast<float4_t *>(s_loc)[tid] = nloc;
            if (tid < BT_NQ) reinterpret_cast<float4_t *>(s_aw)[tid] = naw;
            if (l + 1 < L) {
                if (lthr) nloc = *reinterpret_cast<const float4_t *>(loc + (lq_pair * L + l + 1) * (PT * 2) + (tid & 1) * 4);
                if (tid < BT_NQ) naw = *reinterpret_cast<const float4_t *>(attw + (aq_pair * L + l + 1) * PT);
            }
            __syncthreads();
            BT_TICK(1)   // two barriers around the hand-over of this level's locations / weights

            // ---- A: this lane's point (kpt) of each of its 4 queries; exact bounding window of all corners ----
            float him[BT_NPASS], wim[BT_NPASS], awp[BT_NPASS];
            int hlo[BT_NPASS], wlo[BT_NPASS], okp[BT_NPASS];
            int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) {
                const int slot = p * BT_QPP + slot0;
                const float2_t xy = s_loc[slot * PT + kpt];
                awp[p] = s_aw[slot * PT + kpt];
                const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
                him[p] = sp.h_im; wim[p] = sp.w_im; hlo[p] = sp.h_low; wlo[p] = sp.w_low;
                okp[p] = (sp.ok && qok[p] && H > 0 && W > 0) ? 1 : 0;   // (empty level: no corner inside, nothing to do)
                if (okp[p]) {
                    const int h0 = min(max(sp.h_low, 0), H - 1), h1 = min(max(sp.h_low + 1, 0), H - 1);
                    const int x0 = min(max(sp.w_low, 0), W - 1), x1 = min(max(sp.w_low + 1, 0), W - 1);
                    ymin = min(ymin, h0); ymax = max(ymax, h1); xmin = min(xmin, x0); xmax = max(xmax, x1);
                }
            }
            int r0 = ymin, r1 = -ymax, r2 = xmin, r3 = -xmax;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                r0 = min(r0, __shfl_xor(r0, o)); r1 = min(r1, __shfl_xor(r1, o));
                r2 = min(r2, __shfl_xor(r2, o)); r3 = min(r3, __shfl_xor(r3, o));
            }
            if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; }
            BT_TICK(2)   // A: points + window reduction inside the wave
            __syncthreads();
            const int y0 = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
            const int y1 = -min(min(s_red[0][1], s_red[1][1]), min(s_red[2][1], s_red[3][1]));
            const int x0w = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
            const int x1w = -min(min(s_red[0][3], s_red[1][3]), min(s_red[2][3], s_red[3][3]));
            if (y1 < 0) continue;   // no accepted point at this level (block-uniform): all three gradients stay zero
            const int wh = y1 - y0 + 1, ww = x1w - x0w + 1;
            const int npix = wh * ww;
            const bool use_lds = npix <= BT_WIN;   // block-uniform

            // ---- B: clear the accumulation window ----
            if (use_lds && !(BT_ABL & 32)) {
                for (int i = tid; i < npix * 8; i += BT_THREADS) reinterpret_cast<float4_t *>(gwin)[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            BT_TICK(3)   // window barrier + clear + barrier

            // ---- C: per (query, point): corner reads, the two per-point gradients, grad_value into the window ----
            float *gvl = grad_value + lbase + sub * 4;
#define BT_POINT(K)                                                                                                    \
    {                                                                                                                  \
        const int hl = qbc<K>(hlo[p]), wl = qbc<K>(wlo[p]);                                                            \
        const bool pok = qbc<K>(okp[p]) != 0;                                                                          \
        const float bh = qbc<K>(him[p]), bw = qbc<K>(wim[p]), aw = qbc<K>(awp[p]);                                     \
        const float lh = bh - (float)hl, lw = bw - (float)wl;                                                          \
        const float hh = 1.f - lh, hw = 1.f - lw;                                                                      \
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                            \
        const bool k1 = pok && hl >= 0 && wl >= 0, k2 = pok && hl >= 0 && wl + 1 <= W - 1;                             \
        const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0, k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;             \
        const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);                                        \
        const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);                                        \
        const long g1 = ((long)h0 * W + x0) * MD, g2 = ((long)h0 * W + x1) * MD;                                       \
        const long g3 = ((long)h1 * W + x0) * MD, g4 = ((long)h1 * W + x1) * MD;                                       \
        const float4_t zz = {0.f, 0.f, 0.f, 0.f};                                                                     \
        const float4_t v1 = (BT_ABL & 8) ? zz : *reinterpret_cast<const float4_t *>(vl + g1);                          \
        const float4_t v2 = (BT_ABL & 8) ? zz : *reinterpret_cast<const float4_t *>(vl + g2);                          \
        const float4_t v3 = (BT_ABL & 8) ? zz : *reinterpret_cast<const float4_t *>(vl + g3);                          \
        const float4_t v4 = (BT_ABL & 8) ? zz : *reinterpret_cast<const float4_t *>(vl + g4);                          \
        /* window element offsets (LDS path) / global pointers (fallback path) of the four corners */                 \
        const int e1 = ((h0 - y0) * ww + (x0 - x0w)) * 32 + sub * 4, e2 = ((h0 - y0) * ww + (x1 - x0w)) * 32 + sub * 4; \
        const int e3 = ((h1 - y0) * ww + (x0 - x0w)) * 32 + sub * 4, e4 = ((h1 - y0) * ww + (x1 - x0w)) * 32 + sub * 4; \
        float *d1 = gvl + g1, *d2 = gvl + g2, *d3 = gvl + g3, *d4 = gvl + g4;                                          \
        float g_aw = 0.f, g_x = 0.f, g_y = 0.f;                                                                        \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                \
            const float top = go[p][c], tgv = top * aw;                                                                \
            const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f, a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;    \
            const float ghw = -hw * a1 - lw * a2 + hw * a3 + lw * a4;                                                  \
            const float gww = -hh * a1 + hh * a2 - lh * a3 + lh * a4;                                                  \
            if (use_lds) {                                                                                             \
                if (BT_ABL & 4) { g_aw += w1 * tgv; } else                                                            \
                if (k1) lds_add(e1 + c, w1 * tgv);                                                                     \
                if (k2) lds_add(e2 + c, w2 * tgv);                                                                     \
                if (k3) lds_add(e3 + c, w3 * tgv);                                                                     \
                if (k4) lds_add(e4 + c, w4 * tgv);                                                                     \
            } else if (!(BT_ABL & 2)) {                                                                              \
                if (k1) unsafeAtomicAdd(d1 + c, w1 * tgv);                                                             \
                if (k2) unsafeAtomicAdd(d2 + c, w2 * tgv);                                                             \
                if (k3) unsafeAtomicAdd(d3 + c, w3 * tgv);                                                             \
                if (k4) unsafeAtomicAdd(d4 + c, w4 * tgv);                                                             \
            }                                                                                                          \
            const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;                                                   \
            g_aw += top * val;                                                                                         \
            g_x += (float)W * gww * tgv;                                                                               \
            g_y += (float)H * ghw * tgv;                                                                               \
        }                                                                                                              \
        _Pragma("unroll") for (int o = 4; o > 0; o >>= 1) {                                                            \
            g_aw += __shfl_xor(g_aw, o); g_x += __shfl_xor(g_x, o); g_y += __shfl_xor(g_y, o);                         \
        }                                                                                                              \
        if (sub == 0 && pok && !(BT_ABL & 16)) {   /* rejected points keep the caller's zero fill */                                     \
            const long pi = (qidx[p] * L + l) * PT + K;                                                                \
            grad_attw[pi] = g_aw;                                                                                      \
            grad_loc[2 * pi] = g_x;                                                                                    \
            grad_loc[2 * pi + 1] = g_y;                                                                                \
        }                                                                                                              \
    }
            __attribute__((address_space(3))) float *gwin3 = (__attribute__((address_space(3))) float *)gwin;
            auto lds_add = [&](int e, float v) {   // ds_add_f32 (no return value)
                __hip_atomic_fetch_add(gwin3 + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) {
                BT_POINT(0) BT_POINT(1) BT_POINT(2) BT_POINT(3)
            }
#undef BT_POINT
            BT_TICK(4)   // C: corner reads, gradients, LDS adds (or direct atomics)
            if (!use_lds) continue;   // block-uniform
            __syncthreads();
            BT_TICK(5)   // barrier behind C

            // ---- D: flush the window: one atomic per (pixel, channel), 128 contiguous bytes per pixel ----
            float *gflush = grad_value + lbase;
            const unsigned ww_magic = (1u << 20) / (unsigned)ww + 1u;   // pix / ww exact for pix * ww < 2^20
            for (int i = tid; i < ((BT_ABL & 64) ? 0 : npix * 8); i += BT_THREADS) {
                const int pix = i >> 3, c4 = (i & 7) * 4;
                const int wy = (int)(((unsigned)pix * ww_magic) >> 20), wx = pix - wy * ww;
                const float4_t v = reinterpret_cast<const float4_t *>(gwin)[i];
                float *g = gflush + ((long)(y0 + wy) * W + (x0w + wx)) * MD + c4;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (v[c] != 0.f && !(BT_ABL & 1)) unsafeAtomicAdd(g + c, v[c]);
            }
            BT_TICK(6)   // D: flush
#ifdef BT_PROF
            pacc[8] += 1; pacc[9] += (unsigned)npix;
#endif
        }
    }
#ifdef BT_PROF
    if (tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_bt_prof[i], (unsigned long long)pacc[i]);
#endif
}

}  // namespace

bool msda_bwd_tiled_ok(int D, int L, int P, int Lq, int S, const void *value, const void *grad_out, const void *loc)
{
    return msda_tiled_enabled() && D == 32 && P == 4 && L <= BT_MAXL && Lq == S && Lq >= 4096 && aligned16(value) &&
           aligned16(grad_out) && (reinterpret_cast<uintptr_t>(loc) & 15u) == 0;
}

int msda_bwd_tiled_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                          const float *grad_out, int B, int S, int M, int L, int Lq, float *gv, float *gl, float *gw,
                          hipStream_t st)
{
    const int cus = device_cus();
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BT_LDS);
    }
    const int grid = (cus / 8) * 8 * 3;   // persistent: 3 blocks per CU
    VLLM_LAUNCH(msda_bwd_tiled_kernel, dim3(grid), dim3(BT_THREADS), BT_LDS, st, value, shapes, lsi, loc, attw, grad_out, B, S, M, L,
                Lq, gv, gl, gw);
    VLLM_CHECK_LAUNCH("msda_bwd_tiled_kernel");
    return VLLM_OK;
}

#ifdef BT_ABL_ENTRY
extern "C" int bt_abl_run(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                          const float *grad_out, int B, int S, int M, int L, int Lq, float *gv, float *gl, float *gw, void *stream)
{
    return msda_bwd_tiled_launch(value, shapes, lsi, loc, attw, grad_out, B, S, M, L, Lq, gv, gl, gw, (hipStream_t)stream);
}
void set_error(const char *, ...) {}
int msda_tiled_enabled() { return 1; }
#ifdef BT_PROF
extern "C" int bt_abl_prof(long *out)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bt_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bt_prof), z, sizeof(z));
    return 0;
}
#endif
#endif

}  // namespace vllm
