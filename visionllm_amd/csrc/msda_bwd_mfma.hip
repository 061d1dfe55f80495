// MSDA backward for the encoder self-attention case (queries == pyramid pixels, Lq == S, D = 32, P = 4), generation 2:
// grad_value through the matrix cores.
//
// What the phase clock of msda_bwd_tiled.hip showed (profiles/r03_msda_bwd_phases.txt): 86 % of its 22 ms is the phase that
// accumulates grad_value in LDS with ds_add_f32 -- 65 536 lane-atomics per (query tile, level), and the LDS retires a float
// atomic every ~2 cycles per LANE, whatever the addresses.  But the scatter factors: for a tile of 128 queries and a window
// of pixels
//      grad_value[pix, ch] += sum_q  S[pix, q] * grad_out[q, ch],     S[pix, q] = sum over the 16 (point, corner) pairs of
//                                                                     query q that land on pix of  attention weight x bilinear weight
// -- the channel does not enter S.  So per (tile, level):
//   (1) S^T [128 queries][128 window pixels] is built in LDS with 2 048 lane-atomics (one per (query, point, corner): 32x fewer),
//   (2) a wave multiplies a 32-pixel chunk of it with grad_out [128 x 32] on v_mfma_f32_32x32x2_f32 (exact fp32; 64 steps of
//       k = 2 queries; the B operand -- grad_out of the tile -- stays in 64 registers for all levels and rounds of the item),
//   (3) and adds its 32 x 32 result to grad_value with one atomic per (pixel, channel), a wave instruction covering two whole
//       128-byte pixel rows.
// Windows larger than 128 pixels take several rounds of (1)-(3); beyond 1024 pixels (far-away samples) the level falls
// back to direct global atomics per (point, corner, channel).  grad_sampling_loc and grad_attn_weight are computed as
// before (value corners from global memory / L2, one owner per element, plain stores).
//
// Round 4: the SAME kernel serves the DCNv3 backward for group channels 32 (template flag DCN; dcnv3_im2col_cuda.cuh:86-146, 279-857):
// a DCNv3 call is this operator with ONE value map (the input), heads = groups, queries = output pixels, attention weights = the
// mask, and kh * kw sampling points per query that are run as ceil(kh kw / 4) pseudo-levels of 4 points (3 for the 3 x 3 kernel;
// slots beyond kh kw get a rejected location).  What differs is confined to four places: the geometry set-up, where a point's
// location comes from (offset -> location in the reference's own operation order, dcnv3_im2col_cuda.cuh:319-336, so that a point
// lands in the same cell as in the reference), the scale of the location gradient (offset_scale instead of W / H), and the
// indexing of the three per-point outputs.
//
// Follows ms_deform_im2col_cuda.cuh:87-161 (col2im bilinear: grad_value scatter, grad_sampling_loc, grad_attn_weight) and
// the reductions of :301-360.  Summation order differs from the reference's atomics (as every atomic scatter does); tests
// compare against the oracle with the tolerance of the other backward kernels.
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "dcnv3_geo.hpp"

// Timing-only ablation builds: -DBT_ABL=<mask>.  1: no flush atomics, 2: no direct (large-window) atomics, 4: no S scatter,
// 8: no value corner reads (zeros), 16: no grad_loc / grad_attw stores, 32: no MFMA, 64: no grad_value rounds at all.
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
__device__ unsigned long long g_bm_prof[16];   // phase clock of the diagnostics build (tools/msda_bwd_ablate.sh): ticks of wave 0 of every block
#endif

constexpr int BT_THREADS = 256;
constexpr int BT_QPP = BT_THREADS / 8;   // 32 queries per pass (8 lanes x 4 channels = D 32)
constexpr int BT_MAXL = 8;
#ifndef BT_TILE_W
#define BT_TILE_W 16
#endif
#ifndef BT_BLOCKS_PER_CU
#define BT_BLOCKS_PER_CU 2
#endif
constexpr int BT_TH = 8, BT_TW = BT_TILE_W, BT_NQ = BT_TH * BT_TW, BT_NPASS = BT_NQ / BT_QPP;
constexpr int BT_R = 128;                 // window pixels per round (4 waves x one 32-pixel chunk)
constexpr int BT_RP = BT_R + 1;           // row pitch of S^T [query][pixel] in floats (odd: the scatter's banks spread)
constexpr int BT_MAXWIN = 1024;           // larger windows: direct atomics
constexpr int BT_STAGE = 4 * BT_NQ;            // windows up to this many pixels are staged in LDS (in the S^T buffer, free during phase C) for the corner reads
constexpr size_t BT_LDS_WIN = (size_t)BT_NQ * BT_RP * 4 + 16, BT_LDS_LOC = (size_t)BT_NQ * 4 * 8, BT_LDS_AW = (size_t)BT_NQ * 4 * 4;
constexpr size_t BT_LDS = BT_LDS_WIN + BT_LDS_LOC + BT_LDS_AW;
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int K> __device__ __forceinline__ float qbc(float x)   // value of lane K of this lane's quad
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ int qbc(int x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_add(float x)   // x + (x of the lane CTRL selects)
{
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 8 lanes {8 n .. 8 n + 7}; valid in lane 8 n (quad butterfly: quad_perm [1,0,3,2], [2,3,0,1]; then row_shl:4 brings
// the upper quad's sum down: lane i reads lane i + 4)
__device__ __forceinline__ float sum8(float x) { return dpp_add<0x104>(dpp_add<0x4e>(dpp_add<0xb1>(x))); }

template <bool DCN>
__global__ __launch_bounds__(BT_THREADS, BT_BLOCKS_PER_CU) void msda_bwd_mfma_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out, int B, int S, int M,
    int L, int Lq, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attw, const Dcnv3Geo dq,
    const float dscale)
{
    constexpr int D = 32, PT = 4;
    const int DP = DCN ? dq.kh * dq.kw : 0;   // DCNv3: sampling points per (pixel, group); L = (DP + 3) / 4 pseudo-levels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *st = reinterpret_cast<float *>(smem);                                   // S^T [128 queries][BT_RP]: weight of query q on window pixel p of this round
    float2_t *s_loc = reinterpret_cast<float2_t *>(smem + BT_LDS_WIN);             // [128 queries][4 points]
    float *s_aw = reinterpret_cast<float *>(smem + BT_LDS_WIN + BT_LDS_LOC);       // [128 queries][4 points]
    __shared__ int s_H[BT_MAXL], s_W[BT_MAXL], s_q0[BT_MAXL], s_tc[BT_MAXL + 1];
    __shared__ long s_v0[BT_MAXL];
    __shared__ int s_red[4][4];
    __shared__ int s_geo_ok;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = tid & 7;      // 4-channel chunk of this lane
    const int kpt = tid & 3;      // the sampling point this lane evaluates for its quad
    const int slot0 = tid >> 3;   // query slot inside a pass
    const long MD = (long)M * D;

    if (DCN) {
        if (tid == 0) {
            for (int l = 0; l < L; ++l) { s_H[l] = dq.H; s_W[l] = dq.W; s_q0[l] = 0; s_v0[l] = 0; s_tc[l] = 0; }
            s_tc[L] = ((dq.Ho + BT_TH - 1) / BT_TH) * ((dq.Wo + BT_TW - 1) / BT_TW);
            s_geo_ok = 1;
        }
    } else if (tid == 0) {
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

    for (int i = tid; i < (BT_NQ * BT_RP + 3) / 4; i += BT_THREADS) reinterpret_cast<float4_t *>(smem)[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (long j = blockIdx.x >> 3; j < ipx; j += blocks_per_xcd) {
        const long item = (long)xcd * ipx + j;
        if (item >= n_items) break;
        const int t = (int)(item % n_tiles);
        const long bm = item / n_tiles;
        const int m = (int)(bm % M);
        const long b = bm / M;
        int qH, qW, q0, ty, tx;
        if (DCN) {   // the query grid is the OUTPUT map
            qH = dq.Ho; qW = dq.Wo; q0 = 0;
            const int txn = (qW + BT_TW - 1) / BT_TW;
            ty = t / txn; tx = t - ty * txn;
        } else if (geo) {
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
        // B operand of the grad_value product: grad_out[query 2 s + hi][channel l31], s = 0 .. 63 (zero for slots outside the map);
        // it stays in registers for all levels and rounds of the item
        float gor[BT_NQ / 2];
#pragma unroll
        for (int s2 = 0; s2 < BT_NQ / 2; ++s2) {
            bool ok;
            const long pr = pair_of(2 * s2 + hi, ok);
            const float v = grad_out[pr * D + l31];
            gor[s2] = ok ? v : 0.f;
        }
        bool lq_ok, aq_ok;
        const long lq_pair = pair_of(tid >> 1, lq_ok);
        const long aq_pair = pair_of(tid & (BT_NQ - 1), aq_ok);
        const bool lthr = tid < BT_NQ * 2;
        // locations (x, y) of two points of a query per thread (tid < 256) and the 4 weights of a query (tid < 128) of level l.
        // DCNv3: offsets -> locations in input pixels, the reference's arithmetic (dcnv3_im2col_cuda.cuh:300-334: p0 = centre of the
        // kernel footprint, point (i, j) of the kw x kh grid in w-major order); a slot beyond kh * kw gets (-2, -2): rejected.
        float dp0w = 0.f, dp0h = 0.f;
        if (DCN) {
            const int slot = tid >> 1, y = ty * BT_TH + slot / BT_TW, x = tx * BT_TW + slot % BT_TW;
            const int p0_w = ((dq.dw * (dq.kw - 1)) >> 1) - dq.pw + x * dq.sw, p0_h = ((dq.dh * (dq.kh - 1)) >> 1) - dq.ph + y * dq.sh;
            // (every product rounded on its own -- mul_rn, msda_sample.hpp: the backend would fuse mul + add into one fma, and one ulp
            //  of a location of ~100 pixels is 8e-6 of a pixel: visible in the bilinear weights)
            dp0w = (float)p0_w - mul_rn((float)((dq.dw * (dq.kw - 1)) >> 1), dscale);
            dp0h = (float)p0_h - mul_rn((float)((dq.dh * (dq.kh - 1)) >> 1), dscale);
        }
        auto load_loc = [&](int l) -> float4_t {
            if (!DCN) return *reinterpret_cast<const float4_t *>(loc + (lq_pair * L + l) * (PT * 2) + (tid & 1) * 4);
            float4_t r = {-2.f, -2.f, -2.f, -2.f};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = l * PT + (tid & 1) * 2 + e;
                if (j < DP) {
                    const float2_t o2 = *reinterpret_cast<const float2_t *>(loc + (lq_pair * DP + j) * 2);
                    const int i = j / dq.kh, jj = j - i * dq.kh;
                    r[2 * e] = dp0w + mul_rn((float)(i * dq.dw) + o2.x, dscale);
                    r[2 * e + 1] = dp0h + mul_rn((float)(jj * dq.dh) + o2.y, dscale);
                }
            }
            return r;
        };
        auto load_aw = [&](int l) -> float4_t {
            if (!DCN) return *reinterpret_cast<const float4_t *>(attw + (aq_pair * L + l) * PT);
            float4_t r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (l * PT + e < DP) r[e] = attw[aq_pair * DP + l * PT + e];
            return r;
        };
        float4_t nloc = {0.f, 0.f, 0.f, 0.f};
        if (lthr) nloc = load_loc(0);
        float4_t naw = {0.f, 0.f, 0.f, 0.f};
        if (tid < BT_NQ) naw = load_aw(0);

        BT_TICK(0)   // item set-up: decode, grad_output / first level's locations requested
        for (int l = 0; l < L; ++l) {
            const int H = s_H[l], W = s_W[l];
            const long lbase = (b * (long)S + s_v0[l]) * MD + (long)m * D;   // (batch, level, head) origin, channel 0
            const float *vl = value + lbase + sub * 4;

            __syncthreads();   // previous level / item: every read of s_loc, s_aw, gwin, s_red is finished
            if (lthr) reinterpret_cast<float4_t *>(s_loc)[tid] = nloc;
            if (tid < BT_NQ) reinterpret_cast<float4_t *>(s_aw)[tid] = naw;
            if (l + 1 < L) {
                if (lthr) nloc = load_loc(l + 1);
                if (tid < BT_NQ) naw = load_aw(l + 1);
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
                SamplePoint<float> sp;
                if (DCN) {   // xy is the location in input pixels already; acceptance and floor as dcnv3_im2col_cuda.cuh:335-336, 92-93
                    sp.h_im = xy.y; sp.w_im = xy.x;
                    sp.ok = xy.y > -1.f && xy.x > -1.f && xy.y < (float)H && xy.x < (float)W;
                    sp.h_low = sp.ok ? (int)floorf(xy.y) : 0;
                    sp.w_low = sp.ok ? (int)floorf(xy.x) : 0;
                } else sp = sample_point<float>(xy.x, xy.y, H, W);
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
            if (y1 < 0) {   // no accepted point at this level (block-uniform): MSDA: all three gradients stay zero (the caller's fill)
                if (DCN && sub == 0) {   // DCNv3 writes every slot of grad_offset / grad_mask
#pragma unroll
                    for (int p = 0; p < BT_NPASS; ++p)
                        for (int K = 0; K < PT; ++K)
                            if (qok[p] && l * PT + K < DP) {
                                const long pi = qidx[p] * DP + l * PT + K;
                                grad_attw[pi] = 0.f; grad_loc[2 * pi] = 0.f; grad_loc[2 * pi + 1] = 0.f;
                            }
                }
                continue;
            }
            const int wh = y1 - y0 + 1, ww = x1w - x0w + 1;
            const int npix = wh * ww;
            const bool use_win = npix <= BT_MAXWIN;   // block-uniform
            const bool use_stage = npix <= BT_STAGE;  // block-uniform
            BT_TICK(3)   // window barrier
            // ---- B: the value window of this (batch, level, head) into LDS: 8 pixels (1 KiB) per wave instruction, LDS-DMA.  The
            //      window only holds pixels of the map (its box comes from clamped corners), so there is nothing to zero-fill.
            if (use_stage && !(BT_ABL & 8)) {
                const unsigned ww_m = (1u << 20) / (unsigned)ww + 1u;
                for (int p0 = wave * 8; p0 < npix; p0 += 32) {
                    const int pix = min(p0 + (lane >> 3), npix - 1);
                    const int wy = (int)(((unsigned)pix * ww_m) >> 20), wx = pix - wy * ww;
                    const float *src = value + lbase + ((long)(y0 + wy) * W + (x0w + wx)) * MD + (lane & 7) * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(smem + p0 * 128), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            BT_TICK(8)   // B: window staging

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
        float *d1 = gvl + g1, *d2 = gvl + g2, *d3 = gvl + g3, *d4 = gvl + g4;                                          \
        float g_aw = 0.f, g_x = 0.f, g_y = 0.f;                                                                        \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                \
            const float top = go[p][c], tgv = top * aw;                                                                \
            const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f, a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;    \
            const float ghw = -hw * a1 - lw * a2 + hw * a3 + lw * a4;                                                  \
            const float gww = -hh * a1 + hh * a2 - lh * a3 + lh * a4;                                                  \
            if (!use_win && !(BT_ABL & 2)) {                                                                         \
                if (k1) unsafeAtomicAdd(d1 + c, w1 * tgv);                                                             \
                if (k2) unsafeAtomicAdd(d2 + c, w2 * tgv);                                                             \
                if (k3) unsafeAtomicAdd(d3 + c, w3 * tgv);                                                             \
                if (k4) unsafeAtomicAdd(d4 + c, w4 * tgv);                                                             \
            }                                                                                                          \
            const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;                                                   \
            g_aw += top * val;                                                                                         \
            g_x += (DCN ? dscale : (float)W) * gww * tgv;                                                              \
            g_y += (DCN ? dscale : (float)H) * ghw * tgv;                                                              \
        }                                                                                                              \
        _Pragma("unroll") for (int o = 4; o > 0; o >>= 1) {                                                            \
            g_aw += __shfl_xor(g_aw, o); g_x += __shfl_xor(g_x, o); g_y += __shfl_xor(g_y, o);                         \
        }                                                                                                              \
        if (sub == 0 && (DCN ? (qok[p] && l * PT + K < DP) : pok) && !(BT_ABL & 16)) {   /* MSDA: rejected points keep the caller's zero fill; DCNv3 writes every slot */ \
            const long pi = DCN ? qidx[p] * DP + l * PT + K : (qidx[p] * L + l) * PT + K;                              \
            grad_attw[pi] = pok ? g_aw : 0.f;                                                                          \
            grad_loc[2 * pi] = pok ? g_x : 0.f;                                                                        \
            grad_loc[2 * pi + 1] = pok ? g_y : 0.f;                                                                    \
        }                                                                                                              \
    }
            // Staged window (the common case): the per-point gradients only need the four dot products  d_i = <grad_out, corner i>
            // over the 32 channels:  grad_attw = sum_i w_i d_i,  grad_x = W aw (hh (d2 - d1) + lh (d4 - d3)),
            // grad_y = H aw (hw (d3 - d1) + lw (d4 - d2))  -- 16 multiply-adds per lane and point instead of ~100, the sums over the
            // 8 lanes of a query by DPP adds (quad butterfly, then the upper quad onto the lower one: lane 0 of the 8 owns the result).
#define BT_LEAN_POINT(K)                                                                                               \
    {                                                                                                                  \
        const int hl = qbc<K>(hlo[p]), wl = qbc<K>(wlo[p]);                                                            \
        const bool pok = qbc<K>(okp[p]) != 0;                                                                          \
        const float bh = qbc<K>(him[p]), bw = qbc<K>(wim[p]), aw = qbc<K>(awp[p]);                                     \
        const float lh = bh - (float)hl, lw = bw - (float)wl;                                                          \
        const float hh = 1.f - lh, hw = 1.f - lw;                                                                      \
        const bool u0 = hl >= 0, u1 = hl + 1 <= H - 1, c0 = wl >= 0, c1 = wl + 1 <= W - 1;                             \
        const bool k1 = pok && u0 && c0, k2 = pok && u0 && c1, k3 = pok && u1 && c0, k4 = pok && u1 && c1;            \
        const int r0 = ((hl - y0) * ww + (wl - x0w)) * 128 + sub * 16, rw = ww * 128;                                  \
        const float4_t v1 = *reinterpret_cast<const float4_t *>(smem + (k1 ? r0 : 0));                                 \
        const float4_t v2 = *reinterpret_cast<const float4_t *>(smem + (k2 ? r0 + 128 : 0));                           \
        const float4_t v3 = *reinterpret_cast<const float4_t *>(smem + (k3 ? r0 + rw : 0));                            \
        const float4_t v4 = *reinterpret_cast<const float4_t *>(smem + (k4 ? r0 + rw + 128 : 0));                      \
        const float4_t g = go[p];                                                                                      \
        float d1 = g[0] * v1[0] + g[1] * v1[1] + g[2] * v1[2] + g[3] * v1[3];                                          \
        float d2 = g[0] * v2[0] + g[1] * v2[1] + g[2] * v2[2] + g[3] * v2[3];                                          \
        float d3 = g[0] * v3[0] + g[1] * v3[1] + g[2] * v3[2] + g[3] * v3[3];                                          \
        float d4 = g[0] * v4[0] + g[1] * v4[1] + g[2] * v4[2] + g[3] * v4[3];                                          \
        d1 = k1 ? d1 : 0.f; d2 = k2 ? d2 : 0.f; d3 = k3 ? d3 : 0.f; d4 = k4 ? d4 : 0.f;                                \
        d1 = sum8(d1); d2 = sum8(d2); d3 = sum8(d3); d4 = sum8(d4);                                                    \
        if (sub == 0 && (DCN ? (qok[p] && l * PT + K < DP) : pok) && !(BT_ABL & 16)) {   /* MSDA: rejected points keep the caller's zero fill; DCNv3 writes every slot */ \
            const long pi = DCN ? qidx[p] * DP + l * PT + K : (qidx[p] * L + l) * PT + K;                              \
            grad_attw[pi] = pok ? ((hh * hw) * d1 + (hh * lw) * d2) + ((lh * hw) * d3 + (lh * lw) * d4) : 0.f;         \
            grad_loc[2 * pi] = pok ? (DCN ? dscale : (float)W) * aw * (hh * (d2 - d1) + lh * (d4 - d3)) : 0.f;         \
            grad_loc[2 * pi + 1] = pok ? (DCN ? dscale : (float)H) * aw * (hw * (d3 - d1) + lw * (d4 - d2)) : 0.f;     \
        }                                                                                                              \
    }
            if (use_stage) {
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    BT_LEAN_POINT(0) BT_LEAN_POINT(1) BT_LEAN_POINT(2) BT_LEAN_POINT(3)
                }
            } else {
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    BT_POINT(0) BT_POINT(1) BT_POINT(2) BT_POINT(3)
                }
            }
#undef BT_LEAN_POINT
#undef BT_POINT
            BT_TICK(4)   // C: corner reads, grad_loc / grad_attw (or direct atomics)
            if (!use_win || (BT_ABL & 64)) continue;   // block-uniform

            // ---- D: grad_value of this level: rounds of 128 window pixels ----
            __attribute__((address_space(3))) float *st3 = (__attribute__((address_space(3))) float *)st;
            float *gflush = grad_value + lbase + l31;
            const unsigned ww_magic = (1u << 20) / (unsigned)ww + 1u;   // pix / ww exact for pix * ww < 2^20
            // S^T is zero whenever a round starts: zeroed once per kernel, the staged value window is wiped here, and every round
            // takes its own entries back out after the product ("un-scatter": 16 stores per lane instead of a 66 KB clear).
            __syncthreads();   // phase C's reads of the staged window are finished
            if (use_stage) {
                const int nz = ((npix + 7) & ~7) * 8;   // float4 elements the staging wrote
                for (int i = tid; i < nz; i += BT_THREADS) reinterpret_cast<float4_t *>(st)[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
                __syncthreads();
            }
            BT_TICK(5)   // wipe + barriers
            for (int base = 0; base < npix; base += BT_R) {
                // (1) scatter: lane (query slot, sub) owns point sub & 3 of its slot's queries (both lanes sub and sub + 4 evaluated it in
                // phase A) and two of its four corners -- the upper pair for sub < 4, the lower pair otherwise: every lane of the wave
                // has work, 8 LDS float atomics per lane and round.  (An LDS float atomic instruction costs ~100 cycles of the wave's
                // time whatever its lane count; ordered plain read-add-write turns of the four points were measured no faster.)
                // Corners outside the map or the round are skipped (the un-scatter sends their zero to the row's pad column).
                int sidx[BT_NPASS][2];
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    const int hl = hlo[p] + (sub >> 2), wl = wlo[p];          // this lane's corner row
                    const float lh = him[p] - (float)hlo[p], lw = wim[p] - (float)wl;
                    const float wy_ = (sub >> 2) ? lh : 1.f - lh, aw = awp[p];
                    const bool ur = okp[p] && hl >= 0 && hl <= H - 1, c0 = wl >= 0, c1 = wl + 1 <= W - 1;
                    const int row = (p * BT_QPP + slot0) * BT_RP;
                    const int i1 = (hl - y0) * ww + (wl - x0w) - base, i2 = i1 + 1;
                    sidx[p][0] = row + ((ur && c0 && (unsigned)i1 < (unsigned)BT_R) ? i1 : BT_R);
                    sidx[p][1] = row + ((ur && c1 && (unsigned)i2 < (unsigned)BT_R) ? i2 : BT_R);
                    // (predicated, not redirected: atomics of several lanes on one pad word would serialise)
                    if (!(BT_ABL & 4)) {
                        if (sidx[p][0] != row + BT_R) __hip_atomic_fetch_add(st3 + sidx[p][0], (wy_ * (1.f - lw)) * aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (sidx[p][1] != row + BT_R) __hip_atomic_fetch_add(st3 + sidx[p][1], (wy_ * lw) * aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __syncthreads();
                BT_TICK(6)   // scatter + barrier
                // (2) this wave's 32-pixel chunk x grad_out on the matrix cores, (3) atomics straight from the accumulator layout
                const int p0 = base + wave * 32;
                if (p0 < npix) {   // (wave-uniform)
                    f32x16_t acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const float *ap = st + hi * BT_RP + wave * 32 + l31;
                    if (!(BT_ABL & 32)) {
#pragma unroll
                        for (int s2 = 0; s2 < BT_NQ / 2; ++s2)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2 * BT_RP], gor[s2], acc, 0, 0, 0);
                    }
                    // accumulator register r: pixel (r & 3) + 8 (r >> 2) + 4 hi of the chunk, channel l31
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pix = p0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const int wy = (int)(((unsigned)pix * ww_magic) >> 20), wx = pix - wy * ww;
                        if (pix < npix && acc[r] != 0.f && !(BT_ABL & 1))
                            unsafeAtomicAdd(gflush + ((long)(y0 + wy) * W + (x0w + wx)) * MD, acc[r]);
                    }
                }
                BT_TICK(7)   // MFMA + flush
                __syncthreads();   // every wave has read S^T
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) { st[sidx[p][0]] = 0.f; st[sidx[p][1]] = 0.f; }
                BT_TICK(11)   // barrier + un-scatter
            }
#ifdef BT_PROF
            pacc[9] += 1; pacc[10] += (unsigned)npix;
#endif
        }
    }
#ifdef BT_PROF
    if (tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_bm_prof[i], (unsigned long long)pacc[i]);
#endif
}

}  // namespace

int msda_bwd_mfma_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                         const float *grad_out, int B, int S, int M, int L, int Lq, float *gv, float *gl, float *gw,
                         hipStream_t st)
{
    const int cus = device_cus();
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BT_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BT_LDS);
    }
    const int grid = (cus / 8) * 8 * BT_BLOCKS_PER_CU;   // persistent
    VLLM_LAUNCH(msda_bwd_mfma_kernel<false>, dim3(grid), dim3(BT_THREADS), BT_LDS, st, value, shapes, lsi, loc, attw, grad_out, B, S, M, L,
                Lq, gv, gl, gw, Dcnv3Geo{}, 0.f);
    VLLM_CHECK_LAUNCH("msda_bwd_mfma_kernel");
    return VLLM_OK;
}

// DCNv3 backward, fp32, group channels 32, kh * kw <= 4 * BT_MAXL: the kernel above on the DCNv3 geometry.  grad_input must be
// zero-filled by the caller (it is accumulated); grad_offset / grad_mask are written completely.
bool dcnv3_bwd_mfma_takes(const Dcnv3Geo &q)
{
    return q.C == 32 && q.kh * q.kw >= 1 && q.kh * q.kw <= 4 * BT_MAXL && (long)q.N * q.H * q.W * q.G * q.C < (1L << 40);
}
int dcnv3_bwd_mfma_launch(const float *input, const float *offset, const float *mask, const float *grad_out, const Dcnv3Geo &q,
                          float offset_scale, float *grad_input, float *grad_offset, float *grad_mask, hipStream_t st)
{
    const int cus = device_cus();
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BT_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)BT_LDS);
    }
    const int grid = (cus / 8) * 8 * BT_BLOCKS_PER_CU;
    const int L = (q.kh * q.kw + 3) / 4;
    VLLM_LAUNCH(msda_bwd_mfma_kernel<true>, dim3(grid), dim3(BT_THREADS), BT_LDS, st, input, nullptr, nullptr, offset, mask, grad_out, q.N,
                q.H * q.W, q.G, L, q.Ho * q.Wo, grad_input, grad_offset, grad_mask, q, offset_scale);
    VLLM_CHECK_LAUNCH("msda_bwd_mfma_kernel<dcnv3>");
    return VLLM_OK;
}

#ifdef BT_ABL_ENTRY
extern "C" int bt_abl_run(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                          const float *grad_out, int B, int S, int M, int L, int Lq, float *gv, float *gl, float *gw, void *stream)
{
    return msda_bwd_mfma_launch(value, shapes, lsi, loc, attw, grad_out, B, S, M, L, Lq, gv, gl, gw, (hipStream_t)stream);
}
void set_error(const char *, ...) {}
#ifdef BT_PROF
extern "C" int bt_abl_prof(long *out)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bm_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bm_prof), z, sizeof(z));
    return 0;
}
#endif
#endif

}  // namespace vllm
