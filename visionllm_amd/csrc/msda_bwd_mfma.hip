// MSDA backward for the encoder self-attention case (queries == pyramid pixels, Lq == S, D = 32, P = 4): grad_value through the
// matrix cores.  Follows ms_deform_im2col_cuda.cuh:87-161 (col2im bilinear: grad_value scatter, grad_sampling_loc, grad_attn_weight)
// and the reductions of :301-360.
//
// The scatter factors: for a tile of queries and a window of pixels of one (batch, level, head)
//      grad_value[pix, ch] += sum_q  S[pix, q] * grad_out[q, ch],     S[pix, q] = sum over the 16 (point, corner) pairs of
//                                                                     query q that land on pix of  attention weight x bilinear weight
// -- the channel does not enter S (round 2 accumulated the window with one LDS float atomic per (point, corner, CHANNEL): 22 ms).
//
// A persistent block of 4 waves walks items = (batch, head, tile of 8 x 8 queries); three blocks per CU (33 KiB of LDS, <= 168
// registers).  Per item: grad_out of the tile as the B operand of the product, 32 registers for all levels.  Per level:
//   A  a lane evaluates ITS point of its queries (8 lanes per query: lane K of the lower quad owns point K, of the upper quad point
//      K + 2), locations / weights straight from global memory, requested one level ahead;
//   C  grad_sampling_loc / grad_attn_weight: the 8 lanes of a query (4 channels each) walk its four points, 4 x 16 bytes per step from
//      global memory / L1, the dot products reduce-scattered over the 8 lanes by DPP -- before the window barrier, every wave at its
//      own pace (details at the code);
//   -- the exact bounding window of the tile's corners: wave minima by DPP, one barrier --
//   D  rounds of 128 window pixels: (1) S^T [64 queries][128 pixels] in LDS with one lane-atomic per (query, point, corner),
//      (2) a wave multiplies ITS 32-pixel chunk with grad_out [64 x 32] on v_mfma_f32_32x32x2_f32 (exact fp32), skipping the
//      groups of 8 queries that have no corner in its pixel rows and zeroing the entries as it reads them, (3) and adds the
//      32 x 32 result to grad_value with one atomic per (pixel, channel), a wave instruction covering two whole 128-byte pixel
//      rows at byte offsets from a per-round table.  One barrier per round (two between rounds of one level).
//      Windows beyond 8192 pixels: the (point, corner) entries through a table in LDS and one atomic per (entry, channel).
//
// The SAME kernel serves the DCNv3 backward for group channels 32 (template flag DCN; dcnv3_im2col_cuda.cuh:86-146, 279-857):
// a DCNv3 call is this operator with ONE value map (the input), heads = groups, queries = output pixels, attention weights = the
// mask, and kh * kw sampling points per query that are run as ceil(kh kw / 4) pseudo-levels of 4 points (3 for the 3 x 3 kernel;
// slots beyond kh kw get a rejected location).  What differs is confined to four places: the geometry set-up, where a point's
// location comes from (offset -> location in the reference's own operation order, dcnv3_im2col_cuda.cuh:319-336, so that a point
// lands in the same cell as in the reference), the scale of the location gradient (offset_scale instead of W / H), and the
// indexing of the three per-point outputs.
//
// Summation order differs from the reference's atomics (as every atomic scatter does); tests compare against the oracle with the
// tolerance of the other backward kernels.  History and measurements: NOTES/r05.md section 8, profiles/r05_msda_bwd_diet.txt
// (rounds 3-4: 8 x 16 tiles, 2 blocks per CU, window staged in LDS for phase C, per-level hand-over through LDS: 4.05 ms at
// cfg 4 / B = 8; now 2.41 ms).
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "dcnv3_geo.hpp"

// Round 6: the build switches of round 5 are frozen at the values that won (profiles/r05_msda_bwd_diet.txt has every pass) and
// the timing-only ablation / phase-clock builds are gone; they are in the git history (`git log -- msda_bwd_mfma.hip`).

namespace vllm {
namespace {


constexpr int BT_THREADS = 256;
constexpr int BT_QPP = BT_THREADS / 8;   // 32 queries per pass (8 lanes x 4 channels = D 32)
constexpr int BT_MAXL = 8;
constexpr int BT_TILE_W = 8;          // 8 x 8 query tiles
constexpr int BT_BLOCKS_PER_CU = 3;   // 33 KiB of LDS, 168 registers
constexpr int BT_PRIO = 2;            // wave priority while a wave feeds the matrix core and flushes (its block's other phases, and the other blocks', wait less for the round to end): -2 %
// (a window of up to ~130 pixels is copied into LDS -- LDS-DMA, behind the first round's scatter -- and phase C reads its corners
//  there, after the rounds: 128 B / clock instead of the L1's 64; the grad_value product skips the k-steps -- groups of 8 query
//  slots -- that have no corner in the wave's 32 pixels)
constexpr int BT_TH = 8, BT_TW = BT_TILE_W, BT_NQ = BT_TH * BT_TW, BT_NPASS = BT_NQ / BT_QPP;
constexpr int BT_R = 128;                 // window pixels per round (4 waves x one 32-pixel chunk)
constexpr int BT_RP = BT_R + 1;           // row pitch of S^T [query][pixel] in floats (odd: the scatter's banks spread)
constexpr int BT_MAXWIN = 8192;   // larger windows (a tile of coarse-level queries on a fine map: few points on many pixels): one atomic per (point, corner, channel), two whole pixel rows per wave instruction
constexpr size_t BT_LDS_ST = (size_t)BT_NQ * BT_RP * 4 + 16;   // S^T
constexpr int BT_STG_PX = 150;                 // staged window incl. its guards: 18.75 KiB (3 blocks per CU: 3 x 52 KiB, whatever the allocation granule)
constexpr size_t BT_LDS = BT_LDS_ST + (size_t)BT_STG_PX * 128;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef int int2_t __attribute__((ext_vector_type(2)));

template <int K> __device__ __forceinline__ float qbc(float x)   // value of lane K of this lane's quad
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), K * 0x55, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ int qbc(int x) { return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_add(float x)   // x + (x of the lane CTRL selects)
{
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ float dpp_get(float x)   // x of the lane CTRL selects
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// minimum over the wave, in a scalar register: four DPP steps inside the rows of 16 lanes (quad butterfly, half mirror, mirror), lane 15 of
// rows 0 / 2 onto rows 1 / 3, lane 31 onto the upper half, lane 63 read out -- no LDS permutes (six dependent ds_bpermute per value before)
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_min(int x)
{
    return min(x, __builtin_amdgcn_update_dpp(0x7fffffff, x, CTRL, ROWS, 0xf, false));   // (lanes without a source: the identity)
}
__device__ __forceinline__ int wave_min(int x)
{
    x = dpp_min<0xb1, 0xf>(x); x = dpp_min<0x4e, 0xf>(x); x = dpp_min<0x141, 0xf>(x); x = dpp_min<0x140, 0xf>(x);
    x = dpp_min<0x142, 0xa>(x); x = dpp_min<0x143, 0xc>(x);
    return __builtin_amdgcn_readlane(x, 63);
}
// sum over the 8 lanes {8 n .. 8 n + 7}; valid in lane 8 n (quad butterfly: quad_perm [1,0,3,2], [2,3,0,1]; then row_shl:4 brings
// the upper quad's sum down: lane i reads lane i + 4)
__device__ __forceinline__ float sum8(float x) { return dpp_add<0x104>(dpp_add<0x4e>(dpp_add<0xb1>(x))); }

// block-uniform values that come out of LDS: tell the compiler (scalar registers, scalar arithmetic, scalar address offsets)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long uni(long x)
{
    return (long)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)x >> 32)) << 32) |
                  (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)x));
}

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
    __shared__ int s_H[BT_MAXL], s_W[BT_MAXL], s_q0[BT_MAXL], s_tc[BT_MAXL + 1];
    __shared__ long s_v0[BT_MAXL];
    __shared__ int2_t s_rows[2][BT_NQ / 8];
    __shared__ int s_red[2][4][4];   // (two parities: a wave may start the next level pass while another still reads this one's)
    __shared__ int s_geo_ok;
    __shared__ __attribute__((aligned(16))) unsigned s_poff[BT_R];   // byte offset of the round's window pixels inside the (batch, level, head) slice

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sub = tid & 7;      // 4-channel chunk of this lane
    // the sampling point this lane evaluates: lane K of the LOWER quad of a query's 8 lanes owns point K, lane K of the UPPER quad point
    // (K + 2) & 3 -- one DPP broadcast of quad lane K gives the lower quad point K and the upper quad point K + 2 (phase C)
    const int kpt = (tid + ((tid >> 1) & 2)) & 3;
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
    const bool geo = uni(s_geo_ok) != 0;
    const int n_tiles = uni(geo ? s_tc[L] : (Lq + BT_TW - 1) / BT_TW);
    const unsigned n_items = (unsigned)B * (unsigned)M * (unsigned)n_tiles;   // (< 2^31: the launchers' precondition B M Lq < 2^31)
    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;
    // (tile, (batch, head)) of this block's items: one 32-bit division pair here, then carried (the item index advances by a constant)
    const unsigned step_t = blocks_per_xcd % (unsigned)n_tiles, step_bm = blocks_per_xcd / (unsigned)n_tiles;
    unsigned t_u = (xcd * ipx + (blockIdx.x >> 3)) % (unsigned)n_tiles, bm_u = (xcd * ipx + (blockIdx.x >> 3)) / (unsigned)n_tiles;

    for (int i = tid; i < (BT_NQ * BT_RP + 3) / 4; i += BT_THREADS) reinterpret_cast<float4_t *>(smem)[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    int red_par = 0;
    for (unsigned j = blockIdx.x >> 3; j < ipx; j += blocks_per_xcd) {
        const unsigned item = xcd * ipx + j;
        if (item >= n_items) break;
        const int t = (int)t_u;
        const unsigned bm = bm_u;
        t_u += step_t; bm_u += step_bm;
        if (t_u >= (unsigned)n_tiles) { t_u -= (unsigned)n_tiles; ++bm_u; }
        const int m = (int)(bm % (unsigned)M);
        const long b = (long)(bm / (unsigned)M);
        int qH, qW, q0, ty, tx;
        if (DCN) {   // the query grid is the OUTPUT map
            qH = dq.Ho; qW = dq.Wo; q0 = 0;
            const int txn = (qW + BT_TW - 1) / BT_TW;
            ty = t / txn; tx = t - ty * txn;
        } else if (geo) {
            int lq = 0;
            while (lq + 1 < L && uni(s_tc[lq + 1]) <= t) ++lq;
            qH = uni(s_H[lq]); qW = uni(s_W[lq]); q0 = uni(s_q0[lq]);
            const int txn = (qW + BT_TW - 1) / BT_TW, tl = t - uni(s_tc[lq]);
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
#pragma unroll
        for (int p = 0; p < BT_NPASS; ++p) qidx[p] = pair_of(p * BT_QPP + slot0, qok[p]);
        // (the first level's points are requested FIRST: phase A waits for them, and the loads retire in order -- behind the 34 loads of
        //  grad_out below it waited for all of them)
        // this lane's point (kpt) of its query of pass p at level l: 8 + 4 bytes straight from global memory (the 4 points of a query
        // are one 32-byte / 16-byte piece; lanes sub and sub + 4 read the same words), requested one level ahead.
        // DCNv3: offset -> location in input pixels, the reference's arithmetic (dcnv3_im2col_cuda.cuh:300-334: p0 = centre of the
        // kernel footprint, point (i, j) of the kw x kh grid in w-major order); a slot beyond kh * kw gets (-2, -2): rejected.
        float2_t nloc[BT_NPASS];
        float naw[BT_NPASS];
        auto load_points = [&](int l) {
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) {
                if (!DCN) {
                    nloc[p] = *reinterpret_cast<const float2_t *>(loc + ((qidx[p] * L + l) * PT + kpt) * 2);
                    naw[p] = attw[(qidx[p] * L + l) * PT + kpt];
                } else {
                    const int slot = p * BT_QPP + slot0, y = ty * BT_TH + slot / BT_TW, x = tx * BT_TW + slot % BT_TW;
                    const int p0_w = ((dq.dw * (dq.kw - 1)) >> 1) - dq.pw + x * dq.sw, p0_h = ((dq.dh * (dq.kh - 1)) >> 1) - dq.ph + y * dq.sh;
                    // (every product rounded on its own -- mul_rn, msda_sample.hpp: the backend would fuse mul + add into one fma, and one ulp
                    //  of a location of ~100 pixels is 8e-6 of a pixel: visible in the bilinear weights)
                    const float dp0w = (float)p0_w - mul_rn((float)((dq.dw * (dq.kw - 1)) >> 1), dscale);
                    const float dp0h = (float)p0_h - mul_rn((float)((dq.dh * (dq.kh - 1)) >> 1), dscale);
                    const int j = l * PT + kpt;
                    nloc[p] = (float2_t){-2.f, -2.f};
                    naw[p] = 0.f;
                    if (j < DP) {
                        const float2_t o2 = *reinterpret_cast<const float2_t *>(loc + (qidx[p] * DP + j) * 2);
                        const int i = j / dq.kh, jj = j - i * dq.kh;
                        nloc[p].x = dp0w + mul_rn((float)(i * dq.dw) + o2.x, dscale);
                        nloc[p].y = dp0h + mul_rn((float)(jj * dq.dh) + o2.y, dscale);
                        naw[p] = attw[qidx[p] * DP + j];
                    }
                }
            }
        };
        load_points(0);
        asm volatile("" ::: "memory");   // (keeps the order of the requests)
        float4_t go[BT_NPASS];   // this lane's 4 channels of grad_output of its query slots
#pragma unroll
        for (int p = 0; p < BT_NPASS; ++p) {
            go[p] = *reinterpret_cast<const float4_t *>(grad_out + qidx[p] * D + sub * 4);
            if (!qok[p]) go[p] = (float4_t){0.f, 0.f, 0.f, 0.f};
        }
        // B operand of the grad_value product: grad_out[query 2 s + hi][channel l31], s = 0 .. 63 (zero for slots outside the map);
        // it stays in registers for all levels and rounds of the item
        float gor[BT_NQ / 2];
        // slot 2 s + hi = (row (2 s) / BT_TW, column (2 s) % BT_TW + hi) of the tile: ONE block-uniform base address (a scalar register
        // pair) + a 32-bit byte offset per load = the lane's constant part + a scalar
        const char *gor_base = reinterpret_cast<const char *>(grad_out + ((b * Lq + q0 + (long)(ty * BT_TH) * qW + tx * BT_TW) * M + m) * D);
        const unsigned gor_lane = (unsigned)(hi * (int)MD + l31) * 4u;
        auto load_gor = [&]() {
#pragma unroll
            for (int s2 = 0; s2 < BT_NQ / 2; ++s2) {
                const int dy = (2 * s2) / BT_TW, dx = (2 * s2) % BT_TW;
                const bool ok = ty * BT_TH + dy < qH && tx * BT_TW + dx + hi < qW;
                const unsigned off = (unsigned)((dy * qW + dx) * (int)MD) * 4u + gor_lane;
                const float v = *reinterpret_cast<const float *>(gor_base + (ok ? off : 0u));
                gor[s2] = ok ? v : 0.f;
            }
        };
        load_gor();

        for (int l = 0; l < L; ++l) {
            const int H = uni(s_H[l]), W = uni(s_W[l]);
            const long lbase = (b * (long)S + uni(s_v0[l])) * MD + (long)m * D;   // (batch, level, head) origin, channel 0


            // ---- A: this lane's point (kpt) of each of its 4 queries; exact bounding window of all corners ----
            float him[BT_NPASS], wim[BT_NPASS], awp[BT_NPASS];
            int hlo[BT_NPASS], wlo[BT_NPASS], okp[BT_NPASS];
            int xmin = 0x7fffffff, xmax = -1;
            int rlo[BT_NPASS], rhi[BT_NPASS];   // rows of the corners of this lane's point per pass: (wave, pass) = one group of 8 query slots
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) {
                const float2_t xy = nloc[p];
                awp[p] = naw[p];
                SamplePoint<float> sp;
                if (DCN) {   // xy is the location in input pixels already; acceptance and floor as dcnv3_im2col_cuda.cuh:335-336, 92-93
                    sp.h_im = xy.y; sp.w_im = xy.x;
                    sp.ok = xy.y > -1.f && xy.x > -1.f && xy.y < (float)H && xy.x < (float)W;
                    sp.h_low = sp.ok ? (int)floorf(xy.y) : 0;
                    sp.w_low = sp.ok ? (int)floorf(xy.x) : 0;
                } else sp = sample_point<float>(xy.x, xy.y, H, W);
                him[p] = sp.h_im; wim[p] = sp.w_im; hlo[p] = sp.h_low; wlo[p] = sp.w_low;
                okp[p] = (sp.ok && qok[p] && H > 0 && W > 0) ? 1 : 0;   // (empty level: no corner inside, nothing to do)
                rlo[p] = 0x7fffffff; rhi[p] = -1;
                if (okp[p]) {
                    const int h0 = min(max(sp.h_low, 0), H - 1), h1 = min(max(sp.h_low + 1, 0), H - 1);
                    const int x0 = min(max(sp.w_low, 0), W - 1), x1 = min(max(sp.w_low + 1, 0), W - 1);
                    rlo[p] = h0; rhi[p] = h1; xmin = min(xmin, x0); xmax = max(xmax, x1);
                }
            }
            if (l + 1 < L) load_points(l + 1);
            // ---- C: per (query, point): the four corner reads and the two per-point gradients; run AFTER the rounds of the level (below).
            // The per-point gradients only need the four dot products  d_i = <grad_out, corner i>  over the 32 channels:
            // grad_attw = sum_i w_i d_i,  grad_x = W aw (hh (d2 - d1) + lh (d4 - d3)),  grad_y = H aw (hw (d3 - d1) + lw (d4 - d2)).
            // The 8 lanes of a query (4 channels each) walk its four points in four steps -- the lower quad in the order 0 1 2 3, the upper
            // quad 2 3 0 1 -- and of a point they only need where its corners are (DPP broadcasts from the lane that owns the point):
            // staged window (<= ~125 pixels, copied by LDS-DMA behind the first round's scatter): ONE LDS address, 4 x 16 bytes at
            // 128 B / clock; otherwise four clamped byte offsets and 4 x 16 bytes from global memory / L1 (64 B / clock: the whole
            // phase is L1-bandwidth bound then); 12 packed multiply-adds per step.  The sums over the 8 lanes are a reduce-scatter:
            // one exchange between the quads (row_half_mirror: step t of one quad meets step t + 2 of the other -- the SAME point -- so
            // the lower quad ends up with points 0, 1 and the upper quad with 2, 3: no selects), then a butterfly inside the quad: 24
            // DPP adds per query and level instead of 48.  Lanes 0, 1 of either quad own exactly the points they hold the sums of, and
            // finish with their own fractions, validity and weight: the four points of a query leave as one 16-byte and one 32-byte piece.
            auto phase_c = [&](auto staged_tag, int wy0, int wx0, int www) {
                constexpr bool STG = decltype(staged_tag)::value;
                const char *vb = reinterpret_cast<const char *>(value + lbase) + sub * 16;
                // staged: pixel (wy, wx) of the window at byte BT_LDS_ST + ((www + 1) + wy www + wx) 128.  The guard of www + 1 pixels in
                // front and behind keeps the four corner addresses of every accepted point inside the buffer (row -1 / H, column
                // -1 / W: their sums are discarded below); ONE broadcast per step: the first corner's address
                const int rw = www * 128, sub16 = sub * 16 + (int)BT_LDS_ST + (www + 1) * 128;
#define BT_STEP(T)                                                                                                     \
    {                                                                                                                  \
        float4_t v1, v2, v3, v4;                                                                                       \
        if (STG) {                                                                                                     \
            const int r = qbc<T>(co[0]) + sub16;                                                                       \
            v1 = *reinterpret_cast<const float4_t *>(smem + r); v2 = *reinterpret_cast<const float4_t *>(smem + r + 128); \
            v3 = *reinterpret_cast<const float4_t *>(smem + r + rw); v4 = *reinterpret_cast<const float4_t *>(smem + r + rw + 128); \
        } else {                                                                                                       \
            v1 = *reinterpret_cast<const float4_t *>(vb + (size_t)(unsigned)qbc<T>(co[0]));                            \
            v2 = *reinterpret_cast<const float4_t *>(vb + (size_t)(unsigned)qbc<T>(co[1]));                            \
            v3 = *reinterpret_cast<const float4_t *>(vb + (size_t)(unsigned)qbc<T>(co[2]));                            \
            v4 = *reinterpret_cast<const float4_t *>(vb + (size_t)(unsigned)qbc<T>(co[3]));                            \
        }                                                                                                              \
        const float2_t a1 = g.hi * v1.hi + g.lo * v1.lo, a2 = g.hi * v2.hi + g.lo * v2.lo;                             \
        const float2_t a3 = g.hi * v3.hi + g.lo * v3.lo, a4 = g.hi * v4.hi + g.lo * v4.lo;                             \
        R[T][0] = a1.x + a1.y; R[T][1] = a2.x + a2.y; R[T][2] = a3.x + a3.y; R[T][3] = a4.x + a4.y;                    \
    }
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    const bool pok = okp[p] != 0;
                    const int hl = hlo[p], wl = wlo[p];
                    int co[4];   // (this lane's OWN point: byte offsets of its four corners, clamped into the map, inside the slice; a rejected point reads pixel 0)
                    if (STG) co[0] = pok ? ((hl - wy0) * www + (wl - wx0)) * 128 : 0;
                    else {
                        const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
                        const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);
                        co[0] = pok ? (h0 * W + x0) * (int)MD * 4 : 0; co[1] = pok ? (h0 * W + x1) * (int)MD * 4 : 0;
                        co[2] = pok ? (h1 * W + x0) * (int)MD * 4 : 0; co[3] = pok ? (h1 * W + x1) * (int)MD * 4 : 0;
                    }
                    const float4_t g = go[p];
                    float R[4][4];
                    BT_STEP(0) BT_STEP(1) BT_STEP(2) BT_STEP(3)
                    float kd[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        // step j of this quad + step j + 2 of the other one (lane i <-> lane 7 - i of the query's 8), then the quad butterfly
                        const float n0 = dpp_add<0xb1>(dpp_add<0x4e>(R[0][c] + dpp_get<0x141>(R[2][c])));
                        const float n1 = dpp_add<0xb1>(dpp_add<0x4e>(R[1][c] + dpp_get<0x141>(R[3][c])));
                        kd[c] = (sub & 1) ? n1 : n0;
                    }
                    const float lh = him[p] - (float)hl, lw = wim[p] - (float)wl, hh = 1.f - lh, hw = 1.f - lw, aw = awp[p];
                    const bool u0 = hl >= 0, u1 = hl + 1 <= H - 1, c0 = wl >= 0, c1 = wl + 1 <= W - 1;
                    const float d1 = (u0 && c0) ? kd[0] : 0.f, d2 = (u0 && c1) ? kd[1] : 0.f, d3 = (u1 && c0) ? kd[2] : 0.f, d4 = (u1 && c1) ? kd[3] : 0.f;
                    // every slot of a real query is written (a rejected point: zeros), so the two per-point gradients need no zero fill by
                    // the caller (vllm_msda_backward_f32_writes_point_grads); DCNv3: kh kw slots per query
                    if ((sub & 2) == 0 && qok[p] && (!DCN || l * PT + kpt < DP)) {
                        const long pi = DCN ? qidx[p] * DP + l * PT + kpt : (qidx[p] * L + l) * PT + kpt;
                        grad_attw[pi] = pok ? ((hh * hw) * d1 + (hh * lw) * d2) + ((lh * hw) * d3 + (lh * lw) * d4) : 0.f;
                        *reinterpret_cast<float2_t *>(grad_loc + 2 * pi) =
                            (float2_t){pok ? (DCN ? dscale : (float)W) * aw * (hh * (d2 - d1) + lh * (d4 - d3)) : 0.f,
                                       pok ? (DCN ? dscale : (float)H) * aw * (hw * (d3 - d1) + lw * (d4 - d2)) : 0.f};
                    }
                }
#undef BT_STEP
            };
            int r2 = xmin, r3 = -xmax;
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) rhi[p] = -rhi[p];
            r2 = wave_min(r2); r3 = wave_min(r3);
#pragma unroll
            for (int p = 0; p < BT_NPASS; ++p) { rlo[p] = wave_min(rlo[p]); rhi[p] = wave_min(rhi[p]); }
            int r0 = rlo[0], r1 = rhi[0];
#pragma unroll
            for (int p = 1; p < BT_NPASS; ++p) { r0 = min(r0, rlo[p]); r1 = min(r1, rhi[p]); }
            int2_t *rows = s_rows[red_par];   // [group of 8 slots] = (first row, -last row) of its corners: which k-steps of the product touch which pixels
            if (lane == 0) {
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) rows[p * 4 + wave] = (int2_t){rlo[p], rhi[p]};
            }
            int (*red)[4] = s_red[red_par];
            red_par ^= 1;
            if (lane == 0) { red[wave][0] = r0; red[wave][1] = r1; red[wave][2] = r2; red[wave][3] = r3; }
            __syncthreads();   // (also: every wave is done with the previous level pass -- S^T is zero again, the staging buffer has been read)
            const int y0 = uni(min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0])));
            const int y1 = -uni(min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1])));
            const int x0w = uni(min(min(red[0][2], red[1][2]), min(red[2][2], red[3][2])));
            const int x1w = -uni(min(min(red[0][3], red[1][3]), min(red[2][3], red[3][3])));
            const int wh = y1 - y0 + 1, ww = x1w - x0w + 1;
            const int npix = wh * ww;
            const bool use_win = npix <= BT_MAXWIN;   // block-uniform
            // phase C from LDS: the window fits the staging buffer with its guards (block-uniform)
            const bool staged = y1 >= 0 && ((npix + 7) & ~7) + 2 * (ww + 1) <= BT_STG_PX;
            bool stage_waited = false;
            do {
            if (y1 < 0) break;   // no accepted point at this level (block-uniform): nothing for grad_value
            if (staged) {
                // the value window of this (batch, level, head) into LDS: 8 pixels (1 KiB) per wave instruction, LDS-DMA; it only holds
                // pixels of the map (its box comes from clamped corners).  Waited for in front of the first round's barrier.
                const unsigned ww_m = (1u << 20) / (unsigned)ww + 1u;
                for (int p0 = wave * 8; p0 < npix; p0 += 32) {
                    const int pix = min(p0 + (lane >> 3), npix - 1);
                    const int wy = (int)(((unsigned)pix * ww_m) >> 20), wx = pix - wy * ww;
                    const float *src = value + lbase + ((long)(y0 + wy) * W + (x0w + wx)) * MD + (lane & 7) * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(smem + BT_LDS_ST + (ww + 1) * 128 + p0 * 128), 16, 0, 0);
                }
            }

            // ---- D: grad_value of this level: rounds of 128 window pixels ----
            __attribute__((address_space(3))) float *st3 = (__attribute__((address_space(3))) float *)st;
            const float ww_rcp = __builtin_amdgcn_rcpf((float)ww);   // pix / ww: quotient estimate (pix < 2^23) + one correction step
            // S^T is zero whenever a round starts: zeroed once per kernel, and a wave zeroes the entries of its 32 columns as it feeds
            // them to the matrix core (the groups it skips have no entry there).
            if (!use_win) {   // block-uniform
                // ---- D, sparse: the (point, corner) entries {byte offset in the slice, weight} through a table in LDS (the S^T buffer),
                //      then wave w adds the entries of ITS queries: lane (hi, l31) = channel l31 of query 2 s + hi -- two whole pixel
                //      rows per atomic instruction.  Entry (slot, point, corner row): lane (slot, sub) owns point kpt and row sub >> 2.
                uint2_t *toff = reinterpret_cast<uint2_t *>(smem);                       // [slot][point][corner row] -> byte offsets of its two corners
                float2_t *twt = reinterpret_cast<float2_t *>(smem + BT_NQ * PT * 2 * 8);   //                            -> their weights
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    const int hl = hlo[p] + (sub >> 2), wl = wlo[p];
                    const float lh = him[p] - (float)hlo[p], lw = wim[p] - (float)wl;
                    const float wy_ = (sub >> 2) ? lh : 1.f - lh, aw = awp[p];
                    const bool ur = okp[p] && hl >= 0 && hl <= H - 1, k0 = ur && wl >= 0, k1 = ur && wl + 1 <= W - 1;
                    const int o0 = (hl * W + wl) * (int)MD * 4;
                    const int ei = ((p * BT_QPP + slot0) * PT + kpt) * 2 + (sub >> 2);
                    toff[ei] = (uint2_t){(unsigned)(k0 ? o0 : 0), (unsigned)(k1 ? o0 + (int)MD * 4 : 0)};
                    twt[ei] = (float2_t){k0 ? (wy_ * (1.f - lw)) * aw : 0.f, k1 ? (wy_ * lw) * aw : 0.f};
                }
                if (staged) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stage_waited = true; }
                __syncthreads();
                char *gsl = reinterpret_cast<char *>(grad_value + lbase) + l31 * 4;
#pragma unroll 1
                for (int j = 0; j < BT_NQ / 8; ++j) {
                    const int slot = 2 * (wave * (BT_NQ / 8) + j) + hi;
                    bool ok;
                    const long pr = pair_of(slot, ok);
                    const float gval = ok ? grad_out[pr * D + l31] : 0.f;
#pragma unroll
                    for (int e8 = 0; e8 < PT * 2; ++e8) {
                        const uint2_t a = toff[slot * (PT * 2) + e8];
                        const float2_t w = twt[slot * (PT * 2) + e8];
                        if (w.x != 0.f) unsafeAtomicAdd(reinterpret_cast<float *>(gsl + (size_t)a.x), w.x * gval);
                        if (w.y != 0.f) unsafeAtomicAdd(reinterpret_cast<float *>(gsl + (size_t)a.y), w.y * gval);
                    }
                }
                __syncthreads();
                for (int i = tid; i < BT_NQ * PT * 2; i += BT_THREADS) reinterpret_cast<float4_t *>(smem)[i] = (float4_t){0.f, 0.f, 0.f, 0.f};   // S^T is zero again
                break;
            }
            for (int base = 0; base < npix; base += BT_R) {
                // (1) scatter: lane (query slot, sub) owns ITS point (kpt) of its slot's queries and two of the point's four corners -- the
                // upper pair for sub < 4, the lower pair otherwise (the two quads own every (point, corner row) once): every lane of the
                // wave has work, 2 LDS float atomics per lane, pass and round.  (An LDS float atomic instruction costs ~100 cycles of the
                // wave's time whatever its lane count; ordered plain read-add-write turns of the four points were measured no faster.)
                // Corners outside the map or the round are skipped.  First: the round's table of pixel byte offsets for the flush.
                if (tid < BT_R) {
                    const int pix = base + tid;
                    int wy = (int)((float)pix * ww_rcp), wx = pix - wy * ww;
                    if (wx < 0) { --wy; wx += ww; } else if (wx >= ww) { ++wy; wx -= ww; }
                    s_poff[tid] = (unsigned)(((y0 + wy) * W + (x0w + wx)) * (int)MD) * 4u;   // (slices of 4 GiB or more do not take this kernel)
                }
#pragma unroll
                for (int p = 0; p < BT_NPASS; ++p) {
                    const int hl = hlo[p] + (sub >> 2), wl = wlo[p];          // this lane's corner row
                    const float lh = him[p] - (float)hlo[p], lw = wim[p] - (float)wl;
                    const float wy_ = (sub >> 2) ? lh : 1.f - lh, aw = awp[p];
                    const bool ur = okp[p] && hl >= 0 && hl <= H - 1, c0 = wl >= 0, c1 = wl + 1 <= W - 1;
                    const int row = (p * BT_QPP + slot0) * BT_RP;
                    const int i1 = (hl - y0) * ww + (wl - x0w) - base, i2 = i1 + 1;
                    // (predicated, not redirected to a pad word: atomics of several lanes on one word would serialise)
                    if (ur && c0 && (unsigned)i1 < (unsigned)BT_R) __hip_atomic_fetch_add(st3 + row + i1, (wy_ * (1.f - lw)) * aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (ur && c1 && (unsigned)i2 < (unsigned)BT_R) __hip_atomic_fetch_add(st3 + row + i2, (wy_ * lw) * aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (staged && !stage_waited) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stage_waited = true; }   // this wave's part of the window has landed
                __syncthreads();
                // (2) this wave's 32-pixel chunk x grad_out on the matrix cores, (3) atomics straight from the accumulator layout
                const int p0 = base + wave * 32;
                if (BT_PRIO) __builtin_amdgcn_s_setprio(BT_PRIO);
                if (p0 < npix) {   // (wave-uniform)
                    f32x16_t acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    asm volatile("" : "+v"(acc));   // (an accumulator in registers on every path: a constant-zero first product makes the skip paths re-materialise it)
                    const unsigned ap_lds = (unsigned)(uintptr_t)(st3 + hi * BT_RP + wave * 32 + l31);   // (this wave is the only reader of its 32 columns: it zeroes what it has read)
                    {
                        // only the k-steps whose queries have a corner in this chunk's pixel rows: group g of 8 slots (4 k-steps) touches the
                        // linear window pixels [(first row - y0) ww, (last row - y0 + 1) ww)
                        const int2_t rr = rows[lane & (BT_NQ / 8 - 1)];
                        const int glo = (rr.x - y0) * ww, ghi = (-rr.y - y0 + 1) * ww;
                        const bool hit = rr.y <= 0 && glo < p0 + 32 && ghi > p0;
                        const unsigned gmask = (unsigned)__ballot(hit && lane < BT_NQ / 8);
#pragma unroll
                        for (int g8 = 0; g8 < BT_NQ / 8; ++g8) {
                            if (gmask & (1u << g8)) {
                                // the group's four entries in flight together, zeroed behind the reads (LDS operations complete in order:
                                // at most the four writes outstanding = the reads have landed); written out because the scheduler
                                // otherwise serialises read -> wait -> MFMA through one register
                                float sv0, sv1, sv2, sv3;
                                asm volatile("ds_read_b32 %0, %4 offset:%6\n\tds_read_b32 %1, %4 offset:%7\n\tds_read_b32 %2, %4 offset:%8\n\t"
                                             "ds_read_b32 %3, %4 offset:%9\n\tds_write_b32 %4, %5 offset:%6\n\tds_write_b32 %4, %5 offset:%7\n\t"
                                             "ds_write_b32 %4, %5 offset:%8\n\tds_write_b32 %4, %5 offset:%9\n\ts_waitcnt lgkmcnt(4)"
                                             : "=&v"(sv0), "=&v"(sv1), "=&v"(sv2), "=&v"(sv3)
                                             : "v"(ap_lds), "v"(0.f), "i"(2 * (4 * g8 + 0) * BT_RP * 4), "i"(2 * (4 * g8 + 1) * BT_RP * 4),
                                               "i"(2 * (4 * g8 + 2) * BT_RP * 4), "i"(2 * (4 * g8 + 3) * BT_RP * 4)
                                             : "memory");
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv0, gor[4 * g8 + 0], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv1, gor[4 * g8 + 1], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv2, gor[4 * g8 + 2], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv3, gor[4 * g8 + 3], acc, 0, 0, 0);
                            }
                        }
                    }
                    // accumulator register r: pixel (r & 3) + 8 (r >> 2) + 4 hi of the chunk, channel l31.  The pixel's byte offset inside the
                    // (batch, level, head) slice comes from the round's table (one division per pixel and round instead of one per
                    // (pixel, lane)); a column nobody scattered into -- all columns beyond the window -- has an exactly zero sum.
                    char *gfl = reinterpret_cast<char *>(grad_value + lbase);
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const uint4_t po = *reinterpret_cast<const uint4_t *>(&s_poff[wave * 32 + 8 * j4 + 4 * hi]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float a = acc[4 * j4 + i];
                            if (a != 0.f) unsafeAtomicAdd(reinterpret_cast<float *>(gfl + (size_t)(po[i] + (unsigned)l31 * 4u)), a);
                        }
                    }
                }
                if (BT_PRIO) __builtin_amdgcn_s_setprio(0);
                if (base + BT_R < npix) __syncthreads();   // another round: every wave has read (and zeroed) its columns before the next scatter
            }
            } while (0);
            if (staged) {
                if (!stage_waited) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // (no round ran)
                phase_c(std::true_type{}, y0, x0w, ww);
            } else phase_c(std::false_type{}, 0, 0, 0);
        }
    }
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
    return q.C == 32 && q.kh * q.kw >= 1 && q.kh * q.kw <= 4 * BT_MAXL && (long)q.N * q.H * q.W * q.G * q.C < (1L << 40) &&
           (long)q.H * q.W * q.G * q.C * 4 < (1L << 31) &&   // (32-bit byte offsets inside an image)
           (long)q.N * q.G * q.Ho * q.Wo < (1L << 31);        // (32-bit item index)
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


}  // namespace vllm
