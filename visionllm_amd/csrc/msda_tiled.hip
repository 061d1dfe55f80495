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

constexpr int MT_TH = 8, MT_TW = 16;             // query tile (rows x cols of one level)
constexpr int MT_THREADS = 256;
constexpr int MT_NQ = MT_TH * MT_TW;             // 128 queries per block
constexpr int MT_QPP = MT_THREADS / 8;           // 32 queries per pass (8 lanes x 16 B = D 32 fp32)
constexpr int MT_NPASS = MT_NQ / MT_QPP;         // 4
constexpr int MT_WIN_MAX = 504;                  // window budget in pixels (x 128 B) -> 2 blocks per CU
constexpr int MT_MAXL = 8;


template <int PT, bool LDS>
__device__ __forceinline__ void gather_level(float (&acc)[MT_NPASS][4], const long (&qidx)[MT_NPASS],
                                             const bool (&qok)[MT_NPASS], const float *__restrict__ loc,
                                             const float *__restrict__ attw, int l, int L, int H, int W,
                                             const float *__restrict__ vl, long MD, const float *wb, int y0, int y1,
                                             int x0w, int x1w, int ww)
{
#pragma unroll
    for (int p = 0; p < MT_NPASS; ++p) {
        const float *lp = loc + (qidx[p] * L + l) * (PT * 2);
        const float *wp = attw + qidx[p] * L * PT + l * PT;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const float2_t xy = *reinterpret_cast<const float2_t *>(lp + 2 * k);
            const float aw = wp[k];
            const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
            const int hl = sp.h_low, wl = sp.w_low;
            const float lh = sp.h_im - (float)hl, lw = sp.w_im - (float)wl;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            const bool pok = sp.ok && qok[p];
            const bool k1 = pok && hl >= 0 && wl >= 0;
            const bool k2 = pok && hl >= 0 && wl + 1 <= W - 1;
            const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0;
            const bool k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;
            int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
            int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);
            float4_t v1, v2, v3, v4;
            if (LDS) {
                // rejected points / dead queries may lie outside the window: clamp INTO it (values never used)
                h0 = min(max(h0, y0), y1) - y0; h1 = min(max(h1, y0), y1) - y0;
                x0 = min(max(x0, x0w), x1w) - x0w; x1 = min(max(x1, x0w), x1w) - x0w;
                v1 = *reinterpret_cast<const float4_t *>(wb + (h0 * ww + x0) * 32);
                v2 = *reinterpret_cast<const float4_t *>(wb + (h0 * ww + x1) * 32);
                v3 = *reinterpret_cast<const float4_t *>(wb + (h1 * ww + x0) * 32);
                v4 = *reinterpret_cast<const float4_t *>(wb + (h1 * ww + x1) * 32);
            } else {
                v1 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x0) * MD);
                v2 = *reinterpret_cast<const float4_t *>(vl + ((long)h0 * W + x1) * MD);
                v3 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x0) * MD);
                v4 = *reinterpret_cast<const float4_t *>(vl + ((long)h1 * W + x1) * MD);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f;
                const float a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;
                const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;
                acc[p][c] += val * aw;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one query pass at a time: bounds the loads in flight / VGPRs
    }
}

template <int PT>
__global__ __launch_bounds__(MT_THREADS, 2) void msda_fwd_tiled_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out)
{
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [(MT_WIN_MAX + 8) pixels][128 B]
    float *win = reinterpret_cast<float *>(smem);
    __shared__ int s_H[MT_MAXL], s_W[MT_MAXL], s_q0[MT_MAXL], s_tc[MT_MAXL + 1];
    __shared__ long s_v0[MT_MAXL];
    __shared__ int s_red[4][4];
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = tid & 7;                     // 16-byte channel chunk of this lane
    const long MD = (long)M * D;

    // ---- tile table from the device-side shapes ----
    if (tid == 0) {
        long cum = 0;
        int tc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            s_H[l] = H; s_W[l] = W; s_q0[l] = (int)cum; s_v0[l] = (long)lsi[l]; s_tc[l] = tc;
            tc += ((H + MT_TH - 1) / MT_TH) * ((W + MT_TW - 1) / MT_TW);
            cum += (long)H * W;
        }
        s_tc[L] = tc;
        s_geo_ok = (cum == (long)Lq);
    }
    __syncthreads();
    // queries are tiled on the pyramid geometry when it matches Lq; otherwise as one 1 x Lq strip (still exact)
    const bool geo = s_geo_ok != 0;
    const int n_tiles = geo ? s_tc[L] : (Lq + MT_TW - 1) / MT_TW;
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
            const int txn = (qW + MT_TW - 1) / MT_TW, tl = t - s_tc[lq];
            ty = tl / txn; tx = tl - ty * txn;
        } else {
            qH = 1; qW = Lq; q0 = 0; ty = 0; tx = t;
        }

        // my queries (one per pass)
        long qidx[MT_NPASS];
        bool qok[MT_NPASS];
        float acc[MT_NPASS][4];
#pragma unroll
        for (int p = 0; p < MT_NPASS; ++p) {
            const int slot = p * MT_QPP + (tid >> 3);
            const int y = ty * MT_TH + slot / MT_TW, x = tx * MT_TW + slot % MT_TW;
            qok[p] = y < qH && x < qW;
            const long q = q0 + (long)(qok[p] ? y : 0) * qW + (qok[p] ? x : 0);
            qidx[p] = (b * Lq + q) * M + m;   // (b, q, m) pair index
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[p][c] = 0.f;
        }

        for (int l = 0; l < L; ++l) {
            const int H = s_H[l], W = s_W[l];
            const float *vl = value + (b * (long)S + s_v0[l]) * MD + (long)m * D + sub * 4;

            // ---- A: exact bounding window of every corner this block will touch at level l ----
            int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
#pragma unroll
            for (int p = 0; p < MT_NPASS; ++p) {
                const float *lp = loc + (qidx[p] * L + l) * (PT * 2);
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    const float2_t xy = *reinterpret_cast<const float2_t *>(lp + 2 * k);
                    const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
                    if (qok[p] && sp.ok) {
                        const int h0 = min(max(sp.h_low, 0), H - 1), h1 = min(max(sp.h_low + 1, 0), H - 1);
                        const int x0 = min(max(sp.w_low, 0), W - 1), x1 = min(max(sp.w_low + 1, 0), W - 1);
                        ymin = min(ymin, h0); ymax = max(ymax, h1); xmin = min(xmin, x0); xmax = max(xmax, x1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            int r0 = ymin, r1 = -ymax, r2 = xmin, r3 = -xmax;   // four min-reductions
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                r0 = min(r0, __shfl_xor(r0, o)); r1 = min(r1, __shfl_xor(r1, o));
                r2 = min(r2, __shfl_xor(r2, o)); r3 = min(r3, __shfl_xor(r3, o));
            }
            __syncthreads();   // previous level's window reads (and s_red reads) are finished
            if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; }
            __syncthreads();
            const int y0 = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
            const int y1 = -min(min(s_red[0][1], s_red[1][1]), min(s_red[2][1], s_red[3][1]));
            const int x0w = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
            const int x1w = -min(min(s_red[0][3], s_red[1][3]), min(s_red[2][3], s_red[3][3]));
            if (y1 < 0) continue;                        // no accepted point at this level (block-uniform)
            const int wh = y1 - y0 + 1, ww = x1w - x0w + 1;
            const int npix = wh * ww;
            const bool use_lds = npix <= MT_WIN_MAX;     // block-uniform

            // ---- B: stage the window (LDS-DMA, 8 pixels of 128 B per wave instruction) ----
            if (use_lds) {
                for (int i0 = wave * 8; i0 < npix; i0 += 32) {
                    int pix = i0 + (lane >> 3);
                    pix = pix < npix ? pix : npix - 1;
                    const int wy = pix / ww, wx = pix - wy * ww;
                    const float *g = vl + ((long)(y0 + wy) * W + (x0w + wx)) * MD;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                     (__attribute__((address_space(3))) void *)(win + i0 * 32), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }

            // ---- C: gather + accumulate (same arithmetic as msda_fwd_vec_kernel).  Two separate code bodies
            // (window in LDS / global fallback) keep the register pressure of each bounded. ----
            if (use_lds)
                gather_level<PT, true>(acc, qidx, qok, loc, attw, l, L, H, W, vl, MD, win + sub * 4, y0, y1, x0w, x1w, ww);
            else
                gather_level<PT, false>(acc, qidx, qok, loc, attw, l, L, H, W, vl, MD, win + sub * 4, y0, y1, x0w, x1w, ww);
        }
#pragma unroll
        for (int p = 0; p < MT_NPASS; ++p)
            if (qok[p]) {
                float4_t o = {acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                *reinterpret_cast<float4_t *>(out + qidx[p] * D + sub * 4) = o;
            }
        __syncthreads();   // the next item's first staging must not overwrite a window still being read
    }
}

bool msda_tiled_ok(int D, int L, int P, int Lq, int S, const void *value, const void *out, const void *loc)
{
    return msda_tiled_enabled() && D == 32 && P == 4 && L <= MT_MAXL && Lq == S && Lq >= 4096 && aligned16(value) && aligned16(out) &&
           (reinterpret_cast<uintptr_t>(loc) & 7u) == 0;
}

int msda_tiled_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *attw, int B, int S, int M, int L, int Lq, int P, float *out, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const size_t lds = (size_t)(MT_WIN_MAX + 8) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = (cus / 8) * 8 * 2;   // persistent: 2 blocks per CU
    VLLM_LAUNCH((msda_fwd_tiled_kernel<4>), dim3(grid), dim3(MT_THREADS), lds, st, value, shapes, lsi, loc, attw, B, S, M, L,
                Lq, out);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled_kernel");
    (void)P;
    return VLLM_OK;
}

}  // namespace vllm
