// DCNv3 forward, LDS-tiled kernel with a software pipeline across tiles -- the MSDA generation-7
// scheme (msda_tiled7.hip) for ONE value map.
// Reference: visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:31-84 (bilinear sample with zero padding),
// :217-278 (forward kernel: reference point of an output pixel, kernel_w-outer / kernel_h-inner point order, acceptance).
//
// What the phase clock of the two-blocks-per-CU kernel (dcnv3_tiled.hip, profiles/r02_dcnv3_tiled.txt) showed: 40 % of a
// block's time is issuing / waiting for the window DMA, 14 % the box barrier, 12 % point arithmetic that waits for its
// offsets; only the other block of the CU fills those gaps, and the point table costs 18 KB of LDS and two LDS reads per
// point.  Here a block is software-pipelined across its tiles (rounds 2-5: ONE block of 8 waves per CU on 8 x 16 tiles;
// round 6: two blocks of 4 waves on 8 x 8 tiles, see DP_TH below); a tile is a rectangle of output pixels of one (image, group), a quad per pixel:
//   * no table: quad lane k evaluates points k, k + 4, k + 8 of its pixel (from offsets / mask values requested TWO tiles
//     ahead) and keeps {top corner offset, bottom corner offset, 4 weights x mask} in registers; the gather broadcasts them
//     within the quad by DPP;
//   * two window buffers: while tile n is gathered from one, the window of tile n + 1 is DMA'd into the other, one round (8
//     pixels per wave) after every point of the gather, so the LDS-DMA path streams while the LDS / VALU pipes gather;
//   * ONE barrier per tile (boxes of n + 1 complete, window of n landed, buffer of n - 1 free).
// A tile whose window exceeds the buffer is gathered from global memory by the same lanes.  Results equal the gather kernel
// (dcnv3.hip) to fp32 rounding (the mask value is folded into the weights).
#include "common.hpp"
#include "dcnv3_geo.hpp"

namespace vllm {
namespace {

__device__ float g_dp_zero_px[64];   // a (pixel, group) row of zeros (never written)
__device__ unsigned long long g_dp_prof[16];
#define DP_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// Round 6: 8 x 8 tiles, four waves per block, half-size windows (300 pixels at 32 channels) and TWO blocks per CU instead of one block
// of eight waves on 8 x 16 tiles: the same 8 waves per CU and the same registers, but the two blocks drift apart, so one block's
// gather (LDS) runs next to the other's point arithmetic / window geometry (VALU) -- phases that all eight waves of the single block
// went through in step (57 % of its time outside the gather, profiles/r06_dcnv3_pipe2.txt).  Same box: 8 x 168^2 x 640: 673 -> 629 us,
// 84^2 x 1280: 374 -> 323; 8 x 4 tiles at four blocks per CU (144-pixel windows: most tiles no longer fit): 1198 us.
constexpr int DP_TH = 8, DP_TW = 8, DP_NPX = DP_TH * DP_TW, DP_THREADS = DP_NPX * 4, DP_WAVES = DP_THREADS / 64;
constexpr int DP_BPC = 2;   // blocks per CU
constexpr int DP_BIG = 0x3fffffff;

// value of quad lane K (DPP quad_perm broadcast)
template <int K>
__device__ __forceinline__ int dp_qbi(int x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, true); }
template <int K>
__device__ __forceinline__ float dp_qbf(float x) { return __builtin_bit_cast(float, dp_qbi<K>(__builtin_bit_cast(int, x))); }
// a / b for 0 <= a < 2^22, 0 < b < 2^22 with rb = 1.0f / b: float quotient (error < 1) + fix-up -- ~7 VALU instead of the ~30 of
// the generic 32-bit division sequence (three of them per tile in the item decode, two in the DMA set-up)
__device__ __forceinline__ int dp_div(int a, int b, float rb)
{
    int qd = (int)((float)a * rb);
    const int r = a - qd * b;
    qd += (r >= b) ? 1 : 0;
    qd -= (r < 0) ? 1 : 0;
    return qd;
}
template <int CTRL>
__device__ __forceinline__ int dp_min(int v) { return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float2_t dp_fma2(float w, float2_t v, float2_t a) { return __builtin_elementwise_fma((float2_t){w, w}, v, a); }

struct DpPoints {   // this lane's points (p = k, k + 4, k + 8) of one tile
    int top[3], bot[3];          // hot tile: LDS byte offsets of the top-left / bottom-left corner; cold tile: h_low, w_low
    float w1[3], w2[3], w3[3], w4[3];
};

template <int CPG, int WIN, bool PROF>
__global__ __launch_bounds__(DP_THREADS, 2) void dcnv3_fwd_pipe_kernel(const float *__restrict__ in, const float *__restrict__ off,
                                                                        const float *__restrict__ msk, float *__restrict__ out,
                                                                        Dcnv3Geo q, float offset_scale)
{
    constexpr int LPP = CPG / 4;          // DMA: lanes per pixel (16 bytes of channels each)
    constexpr int PXB = CPG * 4;          // bytes of a (pixel, group) row
    constexpr int PPW = 64 / LPP;         // pixels one wave-wide DMA instruction moves
    constexpr int NR = CPG / 16;          // gather: NR x 16 bytes of channels per quad lane (channels k*4.. and 16+k*4..)
    constexpr int WBUF = (WIN + PPW) * PXB;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    // layout: [0, 256) zero strip (rejected points) | window buffer 0 | window buffer 1 | boxes [3][4]
    int *s_box = reinterpret_cast<int *>(smem + 256 + 2 * WBUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = q.kh * q.kw;
    const int tyN = (q.Ho + DP_TH - 1) / DP_TH, txN = (q.Wo + DP_TW - 1) / DP_TW;
    const unsigned tiles = (unsigned)(tyN * txN);
    const unsigned items = (unsigned)q.N * (unsigned)q.G * tiles;
    const long GC = (long)q.G * CPG;
    unsigned pacc[12] = {};   // (dead in the production instantiation)
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;

    for (int i = tid; i < 64; i += DP_THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    if (tid < 12) s_box[tid] = DP_BIG;
    __syncthreads();

    // per-lane constants
    const int k = tid & 3, pix = tid >> 2;                 // quad lane, pixel of the tile
    const int py = pix / DP_TW, px = pix % DP_TW;
    const int so = k * 16;                                  // this lane's 16 bytes inside each 64-byte half of a pixel row
    const int sub = lane % LPP, lpx = lane / LPP;          // DMA roles
    const int p0w_i = ((q.dw * (q.kw - 1)) >> 1) - q.pw, p0h_i = ((q.dh * (q.kh - 1)) >> 1) - q.ph;
    const float cw = dcn_mul_rn<float>((float)((q.dw * (q.kw - 1)) >> 1), offset_scale), ch = dcn_mul_rn<float>((float)((q.dh * (q.kh - 1)) >> 1), offset_scale);
    float pi_[3], pj_[3];                                   // kernel_w outer, kernel_h inner (:246-249)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int p = k + 4 * r, i = p / q.kh, j = p - i * q.kh;
        pi_[r] = (float)(i * q.dw); pj_[r] = (float)(j * q.dh);
    }

    // XCD-aware walk: XCD x (= blockIdx % 8) owns a contiguous range of items = neighbouring tiles of one (image, group) slab
    const unsigned xcd = blockIdx.x & 7, ipx = (items + 7) >> 3, bpx = gridDim.x >> 3;
    const float r_tiles = 1.0f / (float)tiles, r_G = 1.0f / (float)q.G, r_txN = 1.0f / (float)txN;
    struct Tile { int b, g, ty, tx, oy, ox; bool pok, valid; };
    auto decode = [&](unsigned jj) {
        Tile c;
        const unsigned item = xcd * ipx + jj;
        c.valid = jj < ipx && item < items;
        const unsigned it = c.valid ? item : 0u;
        const int bg = dp_div((int)it, (int)tiles, r_tiles), t = (int)it - bg * (int)tiles;
        c.b = __builtin_amdgcn_readfirstlane(dp_div(bg, q.G, r_G));
        c.g = __builtin_amdgcn_readfirstlane(bg - c.b * q.G);
        c.ty = __builtin_amdgcn_readfirstlane(dp_div(t, txN, r_txN));
        c.tx = __builtin_amdgcn_readfirstlane(t - c.ty * txN);
        c.oy = c.ty * DP_TH + py; c.ox = c.tx * DP_TW + px;
        c.pok = c.valid && c.oy < q.Ho && c.ox < q.Wo;
        return c;
    };
    float2_t o2f[3];   // offsets / mask values of the tile AFTER next, in flight
    float wgf[3];
    auto prefetch = [&](const Tile &c) {
        // (32-bit element indices: N * Ho * Wo * G * K * 2 < 2^31 is a host check)
        const unsigned sidx = (unsigned)(((c.b * q.Ho + (c.pok ? c.oy : 0)) * q.Wo + (c.pok ? c.ox : 0)) * q.G + c.g);
        const unsigned e0 = sidx * (unsigned)K + (unsigned)k;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = k + 4 * r;
            o2f[r] = (float2_t){0.f, 0.f}; wgf[r] = 0.f;
            if (p < K && c.pok) {
                o2f[r] = *reinterpret_cast<const float2_t *>(off + (size_t)((e0 + 4u * r) * 2u));
                wgf[r] = msk[(size_t)(e0 + 4u * r)];
            }
        }
    };

    unsigned jj = blockIdx.x >> 3;
    Tile nxt = decode(jj);
    if (!nxt.valid) return;            // (block-uniform) nothing to do
    prefetch(nxt);
    Tile aft = decode(jj + bpx);       // the tile after next: its offsets are requested when next's have been consumed

    bool cv = false;                   // the current tile exists
    Tile cur = nxt;
    DpPoints pc = {}, pn = {};
    bool cur_hot = true;
    int it = 0;                        // index of `next` in this block's sequence (buffer / box selectors)

    // The window DMA in progress (for the tile that is `next` when it starts and `cur`... no: it must have landed before that
    // tile is gathered, i.e. before the barrier of the following iteration): its rounds are issued one at a time between the
    // points of the gather AND between the points of the following iteration's point arithmetic, so that the LDS-DMA path
    // (~35 GB/s per CU: the window of a tile is ~66 KB) streams through both phases.  All of the state is wave-uniform except
    // the lane's window coordinates.
    bool d_on = false;
    int d_i0 = 0, d_npix = 0, d_y0 = 0, d_x0 = 0, d_ww = 1, d_buf = 0, d_wy = 0, d_wx = 0, d_sty = 0, d_stx = 0;
    const float *d_slab = in;
    const int gcb = (int)GC * 4, rowb = q.W * gcb;    // bytes between neighbouring pixels / rows of a slab (< 2^31: host check)
    constexpr int DSTEP = DP_WAVES * PPW;
    auto dma_round = [&]() {
        if (!d_on || d_i0 >= d_npix) return;     // (wave-uniform)
        const int gy = d_y0 + d_wy, gx = d_x0 + d_wx;
        const bool inside = d_i0 + lpx < d_npix && (unsigned)gy < (unsigned)q.H && (unsigned)gx < (unsigned)q.W;
        const unsigned ofs = (unsigned)(gy * rowb + gx * gcb + sub * 16);
        const char *src = inside ? reinterpret_cast<const char *>(d_slab) + ofs : reinterpret_cast<const char *>(g_dp_zero_px) + sub * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + d_buf + d_i0 * PXB), 16, 0, 0);
        d_i0 += DSTEP;
        d_wx += d_stx; d_wy += d_sty;
        if (d_wx >= d_ww) { d_wx -= d_ww; d_wy += 1; }
    };

    for (;;) {
        const bool nv = nxt.valid;
        // ---- S1: next's points (their offsets / mask values were requested two tiles ago) ----
        int hl[3] = {0, 0, 0}, wl[3] = {0, 0, 0};
        bool okp[3] = {false, false, false};
        if (nv) {
            const float p0w = (float)(p0w_i + nxt.ox * q.sw) - cw, p0h = (float)(p0h_i + nxt.oy * q.sh) - ch;
            int ymin = DP_BIG, ynmin = DP_BIG, xmin = DP_BIG, xnmin = DP_BIG;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int p = k + 4 * r;
                pn.w1[r] = pn.w2[r] = pn.w3[r] = pn.w4[r] = 0.f;
                if (p < K && nxt.pok) {
                    const float2_t o2 = o2f[r];
                    const float wgt = wgf[r];
                    const float loc_w = dcn_loc<float>(p0w, pi_[r], o2.x, offset_scale);
                    const float loc_h = dcn_loc<float>(p0h, pj_[r], o2.y, offset_scale);
                    const bool ok = loc_h > -1.f && loc_w > -1.f && loc_h < (float)q.H && loc_w < (float)q.W;
                    if (ok) {   // (a rejected location, possibly NaN / inf, never reaches the integer arithmetic)
                        const int h = (int)floorf(loc_h), w = (int)floorf(loc_w);
                        const float lh = loc_h - (float)h, lw = loc_w - (float)w, hh = 1.f - lh, hw = 1.f - lw;
                        okp[r] = true; hl[r] = h; wl[r] = w;
                        const float ht = hh * wgt, lt = lh * wgt;   // (mask value folded in: 6 multiplies instead of 8)
                        pn.w1[r] = ht * hw; pn.w2[r] = ht * lw; pn.w3[r] = lt * hw; pn.w4[r] = lt * lw;
                        ymin = min(ymin, h); ynmin = min(ynmin, -h); xmin = min(xmin, w); xnmin = min(xnmin, -w);
                    }
                }
                dma_round();   // (the window of the tile that has just become `cur`)
            }
            // row (16 lanes) minima with four DPP steps per value (quad xor 1, xor 2, row_ror 4, row_ror 8), then one LDS integer
            // minimum per row and value -- written as asm: no compiler-inserted vmcnt wait behind the DMA in flight
            ymin = dp_min<0xB1>(ymin); ynmin = dp_min<0xB1>(ynmin); xmin = dp_min<0xB1>(xmin); xnmin = dp_min<0xB1>(xnmin);
            ymin = dp_min<0x4E>(ymin); ynmin = dp_min<0x4E>(ynmin); xmin = dp_min<0x4E>(xmin); xnmin = dp_min<0x4E>(xnmin);
            ymin = dp_min<0x124>(ymin); ynmin = dp_min<0x124>(ynmin); xmin = dp_min<0x124>(xmin); xnmin = dp_min<0x124>(xnmin);
            ymin = dp_min<0x128>(ymin); ynmin = dp_min<0x128>(ynmin); xmin = dp_min<0x128>(xmin); xnmin = dp_min<0x128>(xnmin);
            if ((lane & 15) == 0) {
                const unsigned ba = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int *)(s_box + (it % 3) * 4);
                asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_min_i32 %0, %3 offset:8\n\tds_min_i32 %0, %4 offset:12"
                             : : "v"(ba), "v"(ymin), "v"(ynmin), "v"(xmin), "v"(xnmin) : "memory");
            }
        }
        while (d_on && d_i0 < d_npix) dma_round();   // whatever is left of cur's window
        DP_TICK(1)   // point arithmetic + box reduction (+ interleaved / remaining DMA issue)
        // ---- barrier: next's boxes complete; cur's window landed; the buffer of the tile before cur is free ----
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DP_TICK(2)   // barrier
        // ---- next's window geometry, corner offsets, DMA state; request the offsets of the tile after next ----
        int ny0 = 0, nx0 = 0, nww = 0, nnpix = 0;
        bool nhot = true;
        const int nbuf = 256 + (it & 1) * WBUF;
        if (nv) {
            const int *box = s_box + (it % 3) * 4;
            int b0, b1, b2, b3;
            {   // (asm read: see the ds_min above)
                const unsigned ba = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const int *)box;
                typedef int int4v_t __attribute__((ext_vector_type(4)));
                int4v_t bv;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(ba) : "memory");
                b0 = __builtin_amdgcn_readfirstlane(bv.x); b1 = __builtin_amdgcn_readfirstlane(bv.y);
                b2 = __builtin_amdgcn_readfirstlane(bv.z); b3 = __builtin_amdgcn_readfirstlane(bv.w);
            }
            const bool any = b0 != DP_BIG;
            ny0 = b0; nx0 = b2;
            nww = any ? (-b3 + 1) - b2 + 1 : 0;
            const int nwh = any ? (-b1 + 1) - b0 + 1 : 0;
            nnpix = nww * nwh;
            nhot = nnpix <= WIN;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (nhot) {
                    pn.top[r] = okp[r] ? nbuf + ((hl[r] - ny0) * nww + (wl[r] - nx0)) * PXB : 0;
                    pn.bot[r] = okp[r] ? pn.top[r] + nww * PXB : 0;
                } else {
                    pn.top[r] = okp[r] ? hl[r] : -2;   // cold: (h_low, w_low); rejected: every corner test fails
                    pn.bot[r] = okp[r] ? wl[r] : -2;
                }
            }
        }
        if (tid < 4) s_box[((it + 2) % 3) * 4 + tid] = DP_BIG;   // the box of the tile after next (its minima start after the NEXT barrier)
        if (aft.valid) prefetch(aft);
        DP_TICK(3)   // window geometry + offsets + prefetch issue
        // DMA of next's window: one round (PPW pixels per wave) per call, issued between the points of the gather
        // next's window DMA starts here (the previous one has landed: vmcnt(0) before the barrier)
        d_on = nv && nhot && nnpix > 0;
        if (d_on) {
            d_i0 = wave * PPW; d_npix = nnpix; d_y0 = ny0; d_x0 = nx0; d_ww = nww; d_buf = nbuf;
            d_slab = in + (long)nxt.b * q.H * q.W * GC + (long)nxt.g * CPG;
            const float r_ww = 1.0f / (float)nww;
            d_sty = dp_div(DSTEP, nww, r_ww); d_stx = DSTEP - d_sty * nww;
            const int i = d_i0 + lpx;
            d_wy = dp_div(i, nww, r_ww); d_wx = i - d_wy * nww;
        }

        // ---- gather cur (window of cur, DPP broadcasts from the owner lane of each point) ----
        if (cv) {
            if constexpr (CPG == 32) {
                // PAIR gather -- bank-conflict-free whatever the pixels are.  A ds_read_b128 is served in groups of 16 lanes = 4
                // quads = 4 unrelated pixels; a 128-byte pixel row is half of the 64 banks, so quads reading 64 contiguous bytes of
                // ONE pixel collide whenever their pixels have equal bank parity (2-4 LDS cycles per group, and the LDS pipe is what
                // bounds this gather: 590 KB per tile).  Here lanes (0,1) of a quad read a 32-byte piece of the LEFT pixel of the 2x2
                // footprint and lanes (2,3) the same piece of the RIGHT pixel -- horizontal neighbours have opposite parity, so a
                // quad always covers 8 banks of each half at the position of its piece -- and the 4 quads of a lane group read 4
                // different pieces (rank ^ t, rank = the quad's number in its group {0,3,5,6} / {1,2,4,7} = quad >> 1): 64 distinct
                // banks, every read.  A lane owns 16 channels of ONE column (two weights per point); the columns are added before
                // the store.
                const int pr_p = lane & 1, pr_s = (lane >> 1) & 1, pr_rank = (lane >> 3) & 3;
                float acc[4][4];   // [read t][channel 8 * (rank ^ t) + 4 p + c]
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
                if (cur_hot) {
                    const int lbase = (int)(uintptr_t)(__attribute__((address_space(3))) char *)smem + pr_s * 128 + pr_rank * 32 + pr_p * 16;
#define DP_POINT(R_, LQ)                                                                                          \
    if (LQ + 4 * R_ < K) {                                                                                         \
        const int t0 = dp_qbi<LQ>(pc.top[R_]) + lbase, t1 = t0 ^ 32, t2 = t0 ^ 64, t3 = t0 ^ 96;                  \
        const int b0 = dp_qbi<LQ>(pc.bot[R_]) + lbase, b1 = b0 ^ 32, b2 = b0 ^ 64, b3 = b0 ^ 96;                  \
        float4_t a0, a1, a2, a3, c0, c1, c2, c3;                                                                  \
        /* reads ISSUED here, WAITED for after the DMA round (its address arithmetic and issue stall hide behind the LDS */ \
        /* round trip); the destination registers are named again only by the waiting statement                          */ \
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"   \
                     "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"      \
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)     \
                     : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");        \
        dma_round();                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory"); \
        const float x1 = dp_qbf<LQ>(pc.w1[R_]), x2 = dp_qbf<LQ>(pc.w2[R_]), x3 = dp_qbf<LQ>(pc.w3[R_]),           \
                    x4 = dp_qbf<LQ>(pc.w4[R_]);                                                                   \
        const float eT = pr_s ? x2 : x1, eB = pr_s ? x4 : x3;                                                     \
        _Pragma("unroll") for (int c = 0; c < 4; c += 2) {                                                        \
            float2_t u0 = {acc[0][c], acc[0][c + 1]}, u1 = {acc[1][c], acc[1][c + 1]};                            \
            float2_t u2 = {acc[2][c], acc[2][c + 1]}, u3 = {acc[3][c], acc[3][c + 1]};                            \
            u0 = dp_fma2(eT, (float2_t){a0[c], a0[c + 1]}, u0); u0 = dp_fma2(eB, (float2_t){c0[c], c0[c + 1]}, u0); \
            u1 = dp_fma2(eT, (float2_t){a1[c], a1[c + 1]}, u1); u1 = dp_fma2(eB, (float2_t){c1[c], c1[c + 1]}, u1); \
            u2 = dp_fma2(eT, (float2_t){a2[c], a2[c + 1]}, u2); u2 = dp_fma2(eB, (float2_t){c2[c], c2[c + 1]}, u2); \
            u3 = dp_fma2(eT, (float2_t){a3[c], a3[c + 1]}, u3); u3 = dp_fma2(eB, (float2_t){c3[c], c3[c + 1]}, u3); \
            acc[0][c] = u0.x; acc[0][c + 1] = u0.y; acc[1][c] = u1.x; acc[1][c + 1] = u1.y;                        \
            acc[2][c] = u2.x; acc[2][c + 1] = u2.y; acc[3][c] = u3.x; acc[3][c + 1] = u3.y;                        \
        }                                                                                                         \
        /* pin the sums: otherwise the multiply-adds sink below the DMA round's branch and the loads are spilled */ \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                             \
            asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]), "+v"(acc[t][2]), "+v"(acc[t][3]));                \
    } else {                                                                                                      \
        dma_round();                                                                                              \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
                    if (K == 9) {
                        // Round 6: the nine points of a 3 x 3 kernel as a TWO-DEEP pipeline (msda_tiled9.hip's gather): the eight reads
                        // of point j + 1 are in flight under the sixteen packed multiply-adds of point j -- two register sets, the LDS
                        // returns a wave's reads in order, s_waitcnt lgkmcnt(8) releases the older set.  One point at a time (the
                        // form below, kept for other kernel sizes) pays the LDS round trip of every point behind one DMA round:
                        // 689 -> 664 us at 8 x 168^2 x 640, 383 -> 370 at 84^2 x 1280, same box (profiles/r06_dcnv3_pipe2.txt).
                        // The next point's addresses and this point's weights are evaluated IN FRONT of the wait (pinned).
                        struct DpSet { float4_t a0, a1, a2, a3, c0, c1, c2, c3; };
                        DpSet sa, sb;
#define DP_RD8(SET, T0, B0)                                                                                                   \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"             \
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"                \
                 : "=&v"(SET.a0), "=&v"(SET.a1), "=&v"(SET.a2), "=&v"(SET.a3), "=&v"(SET.c0), "=&v"(SET.c1), "=&v"(SET.c2), "=&v"(SET.c3) \
                 : "v"(T0), "v"(T0 ^ 32), "v"(T0 ^ 64), "v"(T0 ^ 96), "v"(B0), "v"(B0 ^ 32), "v"(B0 ^ 64), "v"(B0 ^ 96));
#define DP_RDP(SET, R_, LQ)                                                                                                   \
    {                                                                                                                         \
        const int t0_ = dp_qbi<LQ>(pc.top[R_]) + lbase, b0_ = dp_qbi<LQ>(pc.bot[R_]) + lbase;                                 \
        DP_RD8(SET, t0_, b0_)                                                                                                 \
    }
#define DP_STEP(SET, R_, LQ, CNT, NX, NR_, NLQ)                                                                               \
    {                                                                                                                         \
        int t0_ = (NX) ? dp_qbi<NLQ>(pc.top[NR_]) + lbase : 0, b0_ = (NX) ? dp_qbi<NLQ>(pc.bot[NR_]) + lbase : 0;             \
        const float x1 = dp_qbf<LQ>(pc.w1[R_]), x2 = dp_qbf<LQ>(pc.w2[R_]), x3 = dp_qbf<LQ>(pc.w3[R_]),                       \
                    x4 = dp_qbf<LQ>(pc.w4[R_]);                                                                               \
        float eT = pr_s ? x2 : x1, eB = pr_s ? x4 : x3;                                                                       \
        asm volatile("" : "+v"(t0_), "+v"(b0_), "+v"(eT), "+v"(eB));                                                          \
        dma_round();                                                                                                          \
        if (CNT) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(SET.a0), "+v"(SET.a1), "+v"(SET.a2), "+v"(SET.a3), "+v"(SET.c0), "+v"(SET.c1), "+v"(SET.c2), "+v"(SET.c3)); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(SET.a0), "+v"(SET.a1), "+v"(SET.a2), "+v"(SET.a3), "+v"(SET.c0), "+v"(SET.c1), "+v"(SET.c2), "+v"(SET.c3)); \
        _Pragma("unroll") for (int c = 0; c < 4; c += 2) {                                                                    \
            float2_t u0 = {acc[0][c], acc[0][c + 1]}, u1 = {acc[1][c], acc[1][c + 1]};                                        \
            float2_t u2 = {acc[2][c], acc[2][c + 1]}, u3 = {acc[3][c], acc[3][c + 1]};                                        \
            u0 = dp_fma2(eT, (float2_t){SET.a0[c], SET.a0[c + 1]}, u0); u0 = dp_fma2(eB, (float2_t){SET.c0[c], SET.c0[c + 1]}, u0); \
            u1 = dp_fma2(eT, (float2_t){SET.a1[c], SET.a1[c + 1]}, u1); u1 = dp_fma2(eB, (float2_t){SET.c1[c], SET.c1[c + 1]}, u1); \
            u2 = dp_fma2(eT, (float2_t){SET.a2[c], SET.a2[c + 1]}, u2); u2 = dp_fma2(eB, (float2_t){SET.c2[c], SET.c2[c + 1]}, u2); \
            u3 = dp_fma2(eT, (float2_t){SET.a3[c], SET.a3[c + 1]}, u3); u3 = dp_fma2(eB, (float2_t){SET.c3[c], SET.c3[c + 1]}, u3); \
            acc[0][c] = u0.x; acc[0][c + 1] = u0.y; acc[1][c] = u1.x; acc[1][c + 1] = u1.y;                                    \
            acc[2][c] = u2.x; acc[2][c + 1] = u2.y; acc[3][c] = u3.x; acc[3][c + 1] = u3.y;                                    \
        }                                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                         \
            asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]), "+v"(acc[t][2]), "+v"(acc[t][3]));                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                    \
        if (NX) { DP_RD8(SET, t0_, b0_) }                                                                                     \
    }
                        DP_RDP(sa, 0, 0) DP_RDP(sb, 0, 1)
                        DP_STEP(sa, 0, 0, 8, true, 0, 2) DP_STEP(sb, 0, 1, 8, true, 0, 3)
                        DP_STEP(sa, 0, 2, 8, true, 1, 0) DP_STEP(sb, 0, 3, 8, true, 1, 1)
                        DP_STEP(sa, 1, 0, 8, true, 1, 2) DP_STEP(sb, 1, 1, 8, true, 1, 3)
                        DP_STEP(sa, 1, 2, 8, true, 2, 0) DP_STEP(sb, 1, 3, 8, false, 0, 0)
                        DP_STEP(sa, 2, 0, 0, false, 0, 0)
#undef DP_STEP
#undef DP_RDP
#undef DP_RD8
                    } else {
                    DP_POINT(0, 0) DP_POINT(0, 1) DP_POINT(0, 2) DP_POINT(0, 3)
                    DP_POINT(1, 0) DP_POINT(1, 1) DP_POINT(1, 2) DP_POINT(1, 3)
                    DP_POINT(2, 0)
                    }
#undef DP_POINT
                } else {
                    // cold tile: this lane's column of the footprint from global memory (clamped addresses, selects decide what
                    // contributes)
                    const float *slab = in + (long)cur.b * q.H * q.W * GC + (long)cur.g * CPG + pr_p * 4;
                    for (int p = 0; p < K; ++p) {
                        const int src = ((lane & ~3) | (p & 3)) << 2, r = p >> 2;
                        const int h = __builtin_amdgcn_ds_bpermute(src, r == 0 ? pc.top[0] : r == 1 ? pc.top[1] : pc.top[2]);
                        const int wl_ = __builtin_amdgcn_ds_bpermute(src, r == 0 ? pc.bot[0] : r == 1 ? pc.bot[1] : pc.bot[2]);
                        // the owner's weights of BOTH columns (the receiver picks its own column)
                        const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w1[0] : r == 1 ? pc.w1[1] : pc.w1[2])));
                        const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w2[0] : r == 1 ? pc.w2[1] : pc.w2[2])));
                        const float o3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w3[0] : r == 1 ? pc.w3[1] : pc.w3[2])));
                        const float o4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w4[0] : r == 1 ? pc.w4[1] : pc.w4[2])));
                        const float eT = pr_s ? o2 : o1, eB = pr_s ? o4 : o3;
                        const int xc = wl_ + pr_s;
                        const bool u0 = h >= 0, u1 = h + 1 >= 0 && h + 1 <= q.H - 1, lx = xc >= 0 && xc <= q.W - 1;
                        const int ya = min(max(h, 0), q.H - 1), yb = min(max(h + 1, 0), q.H - 1), xa = min(max(xc, 0), q.W - 1);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int ch = 8 * (pr_rank ^ t);
                            const float4_t vT = *reinterpret_cast<const float4_t *>(slab + ((long)ya * q.W + xa) * GC + ch);
                            const float4_t vB = *reinterpret_cast<const float4_t *>(slab + ((long)yb * q.W + xa) * GC + ch);
#pragma unroll
                            for (int c = 0; c < 4; ++c)   // selects (not multiplies by 0): a non-finite value at a clamped address must not leak
                                acc[t][c] += eT * ((u0 && lx) ? vT[c] : 0.f) + eB * ((u1 && lx) ? vB[c] : 0.f);
                        }
                        dma_round();
                    }
                }
                DP_TICK(4)   // gather + interleaved DMA issue
                // left + right column: this lane keeps reads (0,1) [pr_s = 0] or (2,3) [pr_s = 1], hands the other two to its
                // partner lane (lane ^ 2) and stores its two 16-byte pieces
                float o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float keep = pr_s ? acc[2 + (c >> 2)][c & 3] : acc[c >> 2][c & 3];
                    const float give = pr_s ? acc[c >> 2][c & 3] : acc[2 + (c >> 2)][c & 3];
                    o[c] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0x4E, 0xf, 0xf, false));
                }
                if (cur.pok) {
                    float *op = out + ((((long)cur.b * q.Ho + cur.oy) * q.Wo + cur.ox) * q.G + cur.g) * CPG + pr_p * 4;
                    const int t0 = pr_s * 2;
                    *reinterpret_cast<float4_t *>(op + 8 * (pr_rank ^ t0)) = (float4_t){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<float4_t *>(op + 8 * (pr_rank ^ (t0 + 1))) = (float4_t){o[4], o[5], o[6], o[7]};
                }
            } else {
            float acc[NR][4];
    #pragma unroll
                for (int h = 0; h < NR; ++h)
    #pragma unroll
                    for (int c = 0; c < 4; ++c) acc[h][c] = 0.f;
                if (cur_hot) {
                    const int lbase = (int)(uintptr_t)(__attribute__((address_space(3))) char *)smem + so;
    #define DP_POINT(R_, LQ)                                                                                          \
        if (LQ + 4 * R_ < K) {                                                                                                            \
            const int t0 = dp_qbi<LQ>(pc.top[R_]) + lbase, t1 = dp_qbi<LQ>(pc.bot[R_]) + lbase;                       \
            float4_t a1, a2, a3, a4, c1, c2, c3, c4;                                                                  \
            /* the reads are ISSUED here and WAITED for after the DMA round: the round's address arithmetic and the issue stall */ \
            /* of its global_load_lds hide behind the LDS round trip.  The destination registers are named again only by the   */ \
            /* waiting statement, so the compiler has no reason to touch them in between.                                      */ \
            if constexpr (NR == 2) {                                                                                  \
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\t"                              \
                             "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:128\n\t"                              \
                             "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:192\n\t"                    \
                             "ds_read_b128 %6, %9 offset:64\n\tds_read_b128 %7, %9 offset:192"                         \
                             : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4) \
                             : "v"(t0), "v"(t1) : "memory");                                                          \
            } else {                                                                                                  \
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\t"                               \
                             "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:64"                                   \
                             : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4) : "v"(t0), "v"(t1) : "memory");             \
                c1 = c2 = c3 = c4 = (float4_t){0.f, 0.f, 0.f, 0.f};                                                   \
            }                                                                                                         \
            dma_round();                                                                                              \
            if constexpr (NR == 2) {                                                                                  \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4) : : "memory"); \
            } else {                                                                                                  \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : : "memory");           \
            }                                                                                                         \
            const float e1 = dp_qbf<LQ>(pc.w1[R_]), e2 = dp_qbf<LQ>(pc.w2[R_]), e3 = dp_qbf<LQ>(pc.w3[R_]),           \
                        e4 = dp_qbf<LQ>(pc.w4[R_]);                                                                   \
            _Pragma("unroll") for (int c = 0; c < 4; c += 2) {                                                        \
                float2_t t = {acc[0][c], acc[0][c + 1]};                                                              \
                t = dp_fma2(e1, (float2_t){a1[c], a1[c + 1]}, t); t = dp_fma2(e2, (float2_t){a2[c], a2[c + 1]}, t);   \
                t = dp_fma2(e3, (float2_t){a3[c], a3[c + 1]}, t); t = dp_fma2(e4, (float2_t){a4[c], a4[c + 1]}, t);   \
                acc[0][c] = t.x; acc[0][c + 1] = t.y;                                                                 \
                if constexpr (NR == 2) {                                                                              \
                    float2_t u = {acc[NR - 1][c], acc[NR - 1][c + 1]};                                                \
                    u = dp_fma2(e1, (float2_t){c1[c], c1[c + 1]}, u); u = dp_fma2(e2, (float2_t){c2[c], c2[c + 1]}, u); \
                    u = dp_fma2(e3, (float2_t){c3[c], c3[c + 1]}, u); u = dp_fma2(e4, (float2_t){c4[c], c4[c + 1]}, u); \
                    acc[NR - 1][c] = u.x; acc[NR - 1][c + 1] = u.y;                                                   \
                }                                                                                                     \
            }                                                                                                         \
            /* pin the sums: otherwise the multiply-adds sink below the DMA round's branch and the loads are spilled */ \
            _Pragma("unroll") for (int h = 0; h < NR; ++h)                                                            \
                asm volatile("" : "+v"(acc[h][0]), "+v"(acc[h][1]), "+v"(acc[h][2]), "+v"(acc[h][3]));                \
        } else {                                                                                                      \
            dma_round();                                                                                              \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);
                    // (the +128 / +192 immediates are the right-hand pixel of the corner pair: PXB = 128 for 32 channels; for 16
                    // channels a pixel is 64 bytes and the pair's second pixel sits at +64)
                    DP_POINT(0, 0) DP_POINT(0, 1) DP_POINT(0, 2) DP_POINT(0, 3)
                    DP_POINT(1, 0) DP_POINT(1, 1) DP_POINT(1, 2) DP_POINT(1, 3)
                    DP_POINT(2, 0)
    #undef DP_POINT
                } else {
                    // cold tile: corners from global memory (clamped addresses, selects decide what contributes)
                    const float *slab = in + (long)cur.b * q.H * q.W * GC + (long)cur.g * CPG + k * 4;
                    for (int p = 0; p < K; ++p) {
                        const int src = ((lane & ~3) | (p & 3)) << 2, r = p >> 2;
                        const int tb_x = __builtin_amdgcn_ds_bpermute(src, r == 0 ? pc.top[0] : r == 1 ? pc.top[1] : pc.top[2]);
                        const int tb_y = __builtin_amdgcn_ds_bpermute(src, r == 0 ? pc.bot[0] : r == 1 ? pc.bot[1] : pc.bot[2]);
                        float w[4];
                        w[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w1[0] : r == 1 ? pc.w1[1] : pc.w1[2])));
                        w[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w2[0] : r == 1 ? pc.w2[1] : pc.w2[2])));
                        w[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w3[0] : r == 1 ? pc.w3[1] : pc.w3[2])));
                        w[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r == 0 ? pc.w4[0] : r == 1 ? pc.w4[1] : pc.w4[2])));
                        const int h = tb_x, wl_ = tb_y;
                        const bool u0 = h >= 0, u1 = h + 1 >= 0 && h + 1 <= q.H - 1, l0 = wl_ >= 0, l1 = wl_ + 1 >= 0 && wl_ + 1 <= q.W - 1;
                        const int ya = min(max(h, 0), q.H - 1), yb = min(max(h + 1, 0), q.H - 1);
                        const int xa = min(max(wl_, 0), q.W - 1), xb = min(max(wl_ + 1, 0), q.W - 1);
    #pragma unroll
                        for (int hh_ = 0; hh_ < NR; ++hh_) {
                            const float4_t a1 = *reinterpret_cast<const float4_t *>(slab + ((long)ya * q.W + xa) * GC + hh_ * 16);
                            const float4_t a2 = *reinterpret_cast<const float4_t *>(slab + ((long)ya * q.W + xb) * GC + hh_ * 16);
                            const float4_t a3 = *reinterpret_cast<const float4_t *>(slab + ((long)yb * q.W + xa) * GC + hh_ * 16);
                            const float4_t a4 = *reinterpret_cast<const float4_t *>(slab + ((long)yb * q.W + xb) * GC + hh_ * 16);
    #pragma unroll
                            for (int c = 0; c < 4; ++c)   // selects (not multiplies by 0): a non-finite value at a clamped address must not leak
                                acc[hh_][c] += w[0] * ((u0 && l0) ? a1[c] : 0.f) + w[1] * ((u0 && l1) ? a2[c] : 0.f) +
                                               w[2] * ((u1 && l0) ? a3[c] : 0.f) + w[3] * ((u1 && l1) ? a4[c] : 0.f);
                        }
                        dma_round();
                    }
                }
                DP_TICK(4)   // gather + interleaved DMA issue
                if (cur.pok) {
                    float *o = out + ((((long)cur.b * q.Ho + cur.oy) * q.Wo + cur.ox) * q.G + cur.g) * CPG + k * 4;
    #pragma unroll
                    for (int h = 0; h < NR; ++h) *reinterpret_cast<float4_t *>(o + h * 16) = (float4_t){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
                }
            }
            if (PROF) pacc[8] += 1;
        }
        DP_TICK(5)   // stores
        if (!nv) break;
        // cur <- next, next <- the tile after next
        cv = true; cur = nxt; pc = pn; cur_hot = nhot;
        nxt = aft; jj += bpx; aft = decode(jj + bpx); ++it;
        DP_TICK(6)   // rotation + decode
    }
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&g_dp_prof[i], (unsigned long long)pacc[i]);
    }
}

template <int CPG, int WIN, bool PROF>
int dp_go(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out, hipStream_t st)
{
    const int cus = device_cus();
    constexpr size_t lds = 256 + 2 * (size_t)(WIN + 64 / (CPG / 4)) * CPG * 4 + 64;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dcnv3_fwd_pipe_kernel<CPG, WIN, PROF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    VLLM_LAUNCH((dcnv3_fwd_pipe_kernel<CPG, WIN, PROF>), dim3((unsigned)((cus / 8) * 8 * DP_BPC)), dim3(DP_THREADS), lds, st, in, off, msk, out,
                q, offset_scale);
    VLLM_CHECK_LAUNCH("dcnv3_fwd_pipe_kernel");
    return VLLM_OK;
}

}  // namespace

bool dcnv3_pipe_ok(const Dcnv3Geo &q)
{
    return (long)q.N * q.G * ((q.Ho + DP_TH - 1) / DP_TH) * ((q.Wo + DP_TW - 1) / DP_TW) < (1L << 22) &&   // (dp_div operand range)
           (long)q.H * q.W * q.G * q.C * 4 < (1L << 31) &&   // 32-bit byte offsets inside one image
           (long)q.N * q.Ho * q.Wo * q.G * q.kh * q.kw * 2 < (1L << 31);   // 32-bit offset / mask element indices
}

int dcnv3_pipe_launch(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out, int prof,
                      hipStream_t st)
{
    if ((long)q.N * q.Ho * q.Wo * q.G == 0) return VLLM_OK;
    constexpr int W32 = 300, W16 = 600;   // window pixels per buffer: 2 x (300 + 8) x 128 B = 77 KiB per block
    if (q.C == 32) return prof ? dp_go<32, W32, true>(in, off, msk, q, offset_scale, out, st) : dp_go<32, W32, false>(in, off, msk, q, offset_scale, out, st);
    return prof ? dp_go<16, W16, true>(in, off, msk, q, offset_scale, out, st) : dp_go<16, W16, false>(in, off, msk, q, offset_scale, out, st);
}

int dcnv3_pipe_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dp_prof), sizeof(h)) != hipSuccess) {
        set_error("dcnv3_pipe_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dp_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

}  // namespace vllm
