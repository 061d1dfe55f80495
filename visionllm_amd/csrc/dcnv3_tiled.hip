// DCNv3 forward, LDS-tiled kernel (the MSDA generation-4 treatment for ONE value map).
// Reference: visionllmv2/model/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:31-84 (bilinear sample with zero padding),
// :217-278 (forward kernel: reference point of an output pixel, kernel_w-outer / kernel_h-inner point order, acceptance).
//
// The gather kernel (dcnv3.hip) reads every corner of every point from global memory: 9 points x 4 corners x 128 B =
// 4.6 KB of gathered lines per (pixel, group) for 364 B of compulsory traffic, and stalls on the L2 line-request rate
// (15 % of the HBM roofline).  Here a block of 4 waves owns an 8 x 8 tile of output pixels of ONE (image, group):
//   A  point arithmetic: thread (pixel = tid % NPX, tid / NPX) evaluates up to 3 of the pixel's K <= 9 points (location,
//      acceptance, floor, the 4 bilinear weights with the mask value folded in) and the block reduces the bounding box
//      of all accepted points (wave shuffles + 4 LDS integer minima);
//   B  the window -- the box plus the +1 row / column of the far corners, INCLUDING the out-of-map ring, whose pixels
//      are DMA'd from a zero line -- goes to LDS with global_load_lds_dwordx4 (a (pixel, group) row of CPG channels is
//      CPG * 4 bytes: 8 or 4 lanes x 16 B), while every thread turns its points into a table entry {offset of the top
//      corner pair, of the bottom pair, 4 weights}; rejected points aim at a two-pixel zero strip;
//   C  gather: CPG / 4 lanes per pixel, 16 bytes of channels each; a point is one 8-byte + one 16-byte table read
//      (broadcast within the pixel's lanes), four ds_read_b128 and 16 multiply-adds.  No per-corner validity logic.
// Two barriers per tile; two blocks per CU overlap one block's DMA with the other's gather.  A tile whose window exceeds
// the LDS budget (large offsets) is gathered from global memory by the same lanes, so correctness never depends on the
// offsets being small.  Results equal the gather kernel to fp32 rounding (the mask value is folded into the weights).
#include "common.hpp"
#include "dcnv3_geo.hpp"

namespace vllm {

int dcnv3_tiled_enabled();   // runtime.cpp

namespace {

__device__ float g_dcn_zero_px[64];
__device__ unsigned long long g_dcn_prof[16];
#define DT_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }
   // a (pixel, group) row of zeros (stays zero: never written)

// tile of output pixels: 8 x 8 = 64 pixels, a quad each = 4 waves; with a 500-pixel window that is 78 KB of LDS = two blocks
// (8 waves) per CU.  Measured on the 336^2 bench shape: 8 x 8 1.81 ms; 8 x 12 (6 waves, 440-pixel window, 12 waves per CU) 2.02 ms.
constexpr int DT_TH = 8, DT_TW = 8, DT_NPX = DT_TH * DT_TW, DT_KMAX = 9, DT_THREADS = DT_NPX * 4, DT_WAVES = DT_THREADS / 64;
constexpr int DT_BIG = 0x3fffffff;

// Point table in LDS: s_wgt[pixel * KMAX + p] = 4 bilinear weights x mask value (0 for a rejected point);
// s_ofs[pixel * KMAX + p] = hot tile: byte offsets (from the block's LDS base) of the top-left / bottom-left corner (a rejected
// point: the zero strip, twice); cold tile: (h_low, w_low) (rejected: (-2, -2): every corner test fails).

template <int CPG, int WIN, bool PROF>
__global__ __launch_bounds__(DT_THREADS, (2 * DT_WAVES + 3) / 4) void dcnv3_fwd_tiled_kernel(const float *__restrict__ in, const float *__restrict__ off,
                                                                         const float *__restrict__ msk, float *__restrict__ out,
                                                                         Dcnv3Geo q, float offset_scale)
{
    constexpr int LPP = CPG / 4;          // DMA: lanes per pixel (16 bytes of channels each)
    constexpr int NR = CPG / 16;          // gather: a quad per pixel, NR x 16 bytes of channels per lane (channels sub*4.. and 16+sub*4..)
    constexpr int PXB = CPG * 4;          // bytes of a (pixel, group) row
    constexpr int PPW = 64 / LPP;         // pixels one wave-wide DMA instruction moves
    extern __shared__ __attribute__((aligned(256))) char smem[];
    char *s_zero = smem;                                        // 2 pixels of zeros (rejected points)
    char *s_win = smem + 256;                                   // WIN (+ PPW slack) pixels
    float4_t *s_wgt = reinterpret_cast<float4_t *>(s_win + (WIN + PPW) * PXB);
    int2 *s_ofs = reinterpret_cast<int2 *>(s_wgt + DT_NPX * DT_KMAX);
    int *s_box = reinterpret_cast<int *>(s_ofs + DT_NPX * DT_KMAX);   // [2][4]: min y, min -y, min x, min -x

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = q.kh * q.kw;
    const int tyN = (q.Ho + DT_TH - 1) / DT_TH, txN = (q.Wo + DT_TW - 1) / DT_TW;
    const long tiles = (long)tyN * txN;
    const long items = (long)q.N * q.G * tiles;
    const long GC = (long)q.G * CPG;

    for (int i = tid; i < 64; i += DT_THREADS) reinterpret_cast<float *>(s_zero)[i] = 0.f;
    if (tid < 8) s_box[tid] = DT_BIG;
    __syncthreads();

    // XCD-aware walk: XCD x (= blockIdx % 8) owns a contiguous range of items = neighbouring tiles of the same (image,
    // group) slab, whose windows overlap, stay in ITS L2
    const int xcd = blockIdx.x & 7;
    const long ipx = (items + 7) >> 3;
    const int bpx = gridDim.x >> 3;
    const int p0w_i = ((q.dw * (q.kw - 1)) >> 1) - q.pw, p0h_i = ((q.dh * (q.kh - 1)) >> 1) - q.ph;
    const float cw = dcn_mul_rn<float>((float)((q.dw * (q.kw - 1)) >> 1), offset_scale), ch = dcn_mul_rn<float>((float)((q.dh * (q.kh - 1)) >> 1), offset_scale);
    int bsel = 0;
    unsigned pacc[8] = {};   // (dead in the production instantiation)
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    // this thread's points: p = tid / NPX + 4 r; kernel_w outer, kernel_h inner (:246-249)
    float pi_[3], pj_[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int p = tid / DT_NPX + 4 * r, i = p / q.kh, j = p - i * q.kh;
        pi_[r] = (float)(i * q.dw); pj_[r] = (float)(j * q.dh);
    }

    // decode of an item + the global loads of this thread's points (offset pair + mask value), issued one tile ahead
    struct Tile { int b, g, ty, tx, oy, ox; bool pok; };
    auto decode = [&](long item) {
        const unsigned it32 = (unsigned)__builtin_amdgcn_readfirstlane((int)item);   // (items < 2^31: host check)
        const unsigned bg = it32 / (unsigned)tiles;
        const int t = (int)(it32 - bg * (unsigned)tiles);
        Tile c;
        c.b = (int)(bg / (unsigned)q.G); c.g = (int)(bg - (unsigned)c.b * (unsigned)q.G);
        c.ty = t / txN; c.tx = t - c.ty * txN;
        const int pixel = tid % DT_NPX;
        c.oy = c.ty * DT_TH + pixel / DT_TW; c.ox = c.tx * DT_TW + pixel % DT_TW;
        c.pok = c.oy < q.Ho && c.ox < q.Wo;
        return c;
    };
    float2_t o2n[3];
    float wgn[3];
    auto prefetch = [&](const Tile &c) {
        const long sidx = (((long)c.b * q.Ho + (c.pok ? c.oy : 0)) * q.Wo + (c.pok ? c.ox : 0)) * q.G + c.g;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = tid / DT_NPX + 4 * r;
            o2n[r] = (float2_t){0.f, 0.f}; wgn[r] = 0.f;
            if (p < K && c.pok) {
                o2n[r] = *reinterpret_cast<const float2_t *>(off + (sidx * K + p) * 2);
                wgn[r] = msk[sidx * K + p];
            }
        }
    };
    long jj = blockIdx.x >> 3;
    if (!(jj < ipx && (long)xcd * ipx + jj < items)) return;   // (block-uniform) nothing to do
    Tile nxt = decode((long)xcd * ipx + jj);
    prefetch(nxt);

    for (; jj < ipx; jj += bpx) {
        const long item = (long)xcd * ipx + jj;
        if (item >= items) break;                                  // block-uniform
        const Tile cur = nxt;
        const int b = cur.b, g = cur.g, ty = cur.ty, tx = cur.tx, oy = cur.oy, ox = cur.ox;
        const bool pok = cur.pok;

        DT_TICK(0)   // loop control, store drain
        // ---- A: this thread's points (their offsets / mask values were requested a tile ago) ----
        const int pixel = tid % DT_NPX, pg = tid / DT_NPX;
        const float p0w = (float)(p0w_i + ox * q.sw) - cw, p0h = (float)(p0h_i + oy * q.sh) - ch;
        float2_t o2c[3];
        float wgc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { o2c[r] = o2n[r]; wgc[r] = wgn[r]; }
        int hl[3], wl[3];
        float w1[3], w2[3], w3[3], w4[3];
        bool okp[3];
        int ymin = DT_BIG, ynmin = DT_BIG, xmin = DT_BIG, xnmin = DT_BIG;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = pg + 4 * r;
            okp[r] = false; hl[r] = wl[r] = 0; w1[r] = w2[r] = w3[r] = w4[r] = 0.f;
            if (p < K && pok) {
                const float2_t o2 = o2c[r];
                const float wgt = wgc[r];
                const float loc_w = dcn_loc<float>(p0w, pi_[r], o2.x, offset_scale);
                const float loc_h = dcn_loc<float>(p0h, pj_[r], o2.y, offset_scale);
                const bool ok = loc_h > -1.f && loc_w > -1.f && loc_h < (float)q.H && loc_w < (float)q.W;
                if (ok) {   // (a rejected location, possibly NaN / inf, never reaches the integer arithmetic)
                    const int h = (int)floorf(loc_h), w = (int)floorf(loc_w);
                    const float lh = loc_h - (float)h, lw = loc_w - (float)w, hh = 1.f - lh, hw = 1.f - lw;
                    okp[r] = true; hl[r] = h; wl[r] = w;
                    w1[r] = hh * hw * wgt; w2[r] = hh * lw * wgt; w3[r] = lh * hw * wgt; w4[r] = lh * lw * wgt;
                    ymin = min(ymin, h); ynmin = min(ynmin, -h); xmin = min(xmin, w); xnmin = min(xnmin, -w);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ymin = min(ymin, __shfl_xor(ymin, o)); ynmin = min(ynmin, __shfl_xor(ynmin, o));
            xmin = min(xmin, __shfl_xor(xmin, o)); xnmin = min(xnmin, __shfl_xor(xnmin, o));
        }
        DT_TICK(1)   // point arithmetic (incl. the wait for the prefetched offsets)
        int *box = s_box + bsel * 4;
        if (lane == 0) { atomicMin(box + 0, ymin); atomicMin(box + 1, ynmin); atomicMin(box + 2, xmin); atomicMin(box + 3, xnmin); }
        __syncthreads();   // B1: boxes complete; every thread has finished the previous tile's gather
        DT_TICK(2)   // box reduction + barrier 1

        const int y0 = box[0], y1 = -box[1] + 1, x0 = box[2], x1 = -box[3] + 1;   // window rows y0..y1, columns x0..x1
        const bool any = y0 != DT_BIG;
        const int ww = any ? x1 - x0 + 1 : 0, wh = any ? y1 - y0 + 1 : 0;
        const int npix = ww * wh;
        const bool hot = npix <= WIN;   // (block-uniform; an empty window is "hot": every point aims at the zero strip)
        if (tid < 4) s_box[(bsel ^ 1) * 4 + tid] = DT_BIG;   // the next tile's box (its minima start after B2)

        // ---- table entries of this thread's points ----
        constexpr int winbase = 256, zerobase = 0;   // (s_win, s_zero relative to smem)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = pg + 4 * r;
            if (p < K) {
                int2 o;
                if (hot) {
                    o.x = okp[r] ? winbase + ((hl[r] - y0) * ww + (wl[r] - x0)) * PXB : zerobase;
                    o.y = okp[r] ? o.x + ww * PXB : zerobase;
                } else {
                    o.x = okp[r] ? hl[r] : -2;
                    o.y = okp[r] ? wl[r] : -2;
                }
                s_ofs[pixel * DT_KMAX + p] = o;
                s_wgt[pixel * DT_KMAX + p] = (float4_t){w1[r], w2[r], w3[r], w4[r]};
            }
        }
        // the next tile's offsets / mask values (they land while this tile is staged and gathered)
        {
            const long nitem = (long)xcd * ipx + jj + bpx;
            if (jj + bpx < ipx && nitem < items) { nxt = decode(nitem); prefetch(nxt); }
        }
        // ---- B: window DMA ----
        const int sub = lane % LPP, lpx = lane / LPP;
        if (hot && npix > 0) {
            const float inv_ww = 1.f / (float)ww;
            const float *slab = in + (long)b * q.H * q.W * GC + (long)g * CPG + sub * 4;
            for (int i0 = wave * PPW; i0 < npix; i0 += DT_WAVES * PPW) {
                const int i = i0 + lpx;
                int wy = (int)(((float)i + 0.5f) * inv_ww);
                const int wx = i - wy * ww;
                const int gy = y0 + wy, gx = x0 + wx;
                const bool inside = i < npix && (unsigned)gy < (unsigned)q.H && (unsigned)gx < (unsigned)q.W;
                const float *src = inside ? slab + ((long)gy * q.W + gx) * GC : g_dcn_zero_px + sub * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(s_win + i0 * PXB), 16, 0, 0);
            }
        }
        DT_TICK(3)   // table, prefetch issue, DMA issue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // B2: window + table complete
        DT_TICK(4)   // DMA wait + barrier 2

        // ---- C: gather: quad `tid >> 2` owns pixel `tid >> 2` of the tile ----
        {
            const int px = tid >> 2, qs = tid & 3;
            const int gy_o = ty * DT_TH + px / DT_TW, gx_o = tx * DT_TW + px % DT_TW;
            float acc[NR][4];
#pragma unroll
            for (int h = 0; h < NR; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[h][c] = 0.f;
            const int2 *eo = s_ofs + px * DT_KMAX;
            const float4_t *ew = s_wgt + px * DT_KMAX;
            if (hot) {
                const int so = qs * 16;
                auto point = [&](int p) {
                    const int2 tb = eo[p];
                    const float4_t w = ew[p];
#pragma unroll
                    for (int h = 0; h < NR; ++h) {
                        const float4_t a1 = *reinterpret_cast<const float4_t *>(smem + tb.x + so + h * 64);
                        const float4_t a2 = *reinterpret_cast<const float4_t *>(smem + tb.x + PXB + so + h * 64);
                        const float4_t a3 = *reinterpret_cast<const float4_t *>(smem + tb.y + so + h * 64);
                        const float4_t a4 = *reinterpret_cast<const float4_t *>(smem + tb.y + PXB + so + h * 64);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[h][c] += w[0] * a1[c] + w[1] * a2[c] + w[2] * a3[c] + w[3] * a4[c];
                    }
                };
                if (K == 9) {   // three points in flight (a full unroll needs more than the 168 registers of 3 waves per SIMD)
                    for (int p3 = 0; p3 < 9; p3 += 3) {
                        point(p3); point(p3 + 1); point(p3 + 2);
                    }
                } else {
                    for (int p = 0; p < K; ++p) point(p);
                }
            } else {
                const float *slab = in + (long)b * q.H * q.W * GC + (long)g * CPG + qs * 4;
                for (int p = 0; p < K; ++p) {
                    const int2 tb = eo[p];
                    const float4_t w = ew[p];
                    const int h = tb.x, wl_ = tb.y;
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
                }
            }
            if (gy_o < q.Ho && gx_o < q.Wo) {
                float *o = out + ((((long)b * q.Ho + gy_o) * q.Wo + gx_o) * q.G + g) * CPG + qs * 4;
#pragma unroll
                for (int h = 0; h < NR; ++h) *reinterpret_cast<float4_t *>(o + h * 16) = (float4_t){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
            }
        }
        DT_TICK(5)   // gather + store issue
        if (PROF) pacc[6] += 1;
        bsel ^= 1;
    }
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&g_dcn_prof[i], (unsigned long long)pacc[i]);
    }
}

template <int CPG, int WIN, bool PROF>
int dt_go(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out, hipStream_t st)
{
    const int cus = device_cus();
    constexpr size_t lds = 256 + (size_t)(WIN + 64 / (CPG / 4)) * CPG * 4 + (size_t)DT_NPX * DT_KMAX * 24 + 32;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dcnv3_fwd_tiled_kernel<CPG, WIN, PROF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const long items = (long)q.N * q.G * ((q.Ho + DT_TH - 1) / DT_TH) * ((q.Wo + DT_TW - 1) / DT_TW);
    long blocks = (long)(cus / 8) * 8 * 2;
    if (blocks > ((items + 7) / 8) * 8) blocks = ((items + 7) / 8) * 8;
    VLLM_LAUNCH((dcnv3_fwd_tiled_kernel<CPG, WIN, PROF>), dim3((unsigned)blocks), dim3(DT_THREADS), lds, st, in, off, msk, out, q,
                offset_scale);
    VLLM_CHECK_LAUNCH("dcnv3_fwd_tiled_kernel");
    return VLLM_OK;
}

}  // namespace

bool dcnv3_tiled_ok(const Dcnv3Geo &q, const float *in, const float *off, const float *msk, const float *out)
{
    return dcnv3_tiled_enabled() && (q.C == 16 || q.C == 32) && q.kh * q.kw <= DT_KMAX && aligned16(in) && aligned16(out) &&
           (reinterpret_cast<uintptr_t>(off) & 7u) == 0 && msk != nullptr && (long)q.N * q.H * q.W * q.G * q.C < (1L << 40) &&
           (long)q.N * q.G * ((q.Ho + DT_TH - 1) / DT_TH) * ((q.Wo + DT_TW - 1) / DT_TW) < (1L << 31);
}

int dcnv3_tiled_launch(const float *in, const float *off, const float *msk, const Dcnv3Geo &q, float offset_scale, float *out,
                       hipStream_t st)
{
    if ((long)q.N * q.Ho * q.Wo * q.G == 0) return VLLM_OK;
    const bool prof = dcnv3_tiled_enabled() == 4;
    if (q.C == 32) return prof ? dt_go<32, 500, true>(in, off, msk, q, offset_scale, out, st) : dt_go<32, 500, false>(in, off, msk, q, offset_scale, out, st);
    return prof ? dt_go<16, 1000, true>(in, off, msk, q, offset_scale, out, st) : dt_go<16, 1000, false>(in, off, msk, q, offset_scale, out, st);
}


int dcnv3_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dcn_prof), sizeof(h)) != hipSuccess) {
        set_error("dcnv3_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dcn_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}

}  // namespace vllm
