// Row normalisations (bf16 in/out, fp32 statistics) for gfx950.  HBM-bound: 16-byte vector loads, one pass over
// memory (rows are cached in registers), wave-shuffle reductions, optional multi-wave rows for wide C.
//
// Replaces:
//   InternRMSNorm / apex FusedRMSNorm   VisionLLMv2/visionllmv2/model/internvit/modeling_intern_vit.py:33-58
//       y = weight * bf16( x * rsqrt(mean(x^2) + eps) )      (fp32 statistics, cast, THEN multiply: kept)
//   QK-RMSNorm over the flattened H*D    :131-134 (applied in place on the q / k column blocks of the qkv buffer)
//   nn.LayerNorm (CLIP pre_layrnorm / layer_norm1/2, vl_bridge LayerNorm)  transformers CLIPEncoderLayer;
//       visionllmv2/model/modeling_visionllmv2.py:166-167
#include "common.hpp"
#include "kernels.hpp"

namespace vllm {

constexpr int NORM_THREADS = 256;
constexpr int NORM_MAX_CHUNKS = 8;  // upper bound of 16-byte chunks cached per lane (template MAXCH picks 1/2/4/8)

// sum over the wave, in every lane (a scalar register): four DPP steps inside the rows of 16 lanes (quad butterfly, half mirror, mirror),
// lane 15 of rows 0 / 2 onto rows 1 / 3, lane 31 onto the upper half, lane 63 read out.  (Rounds 1-4: six dependent ds_bpermute round
// trips per value.)  A fixed tree: deterministic.
template <int CTRL, int ROWS> __device__ __forceinline__ float dpp_sum_step(float x)
{
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = dpp_sum_step<0xb1, 0xf>(v); v = dpp_sum_step<0x4e, 0xf>(v); v = dpp_sum_step<0x141, 0xf>(v); v = dpp_sum_step<0x140, 0xf>(v);
    v = dpp_sum_step<0x142, 0xa>(v); v = dpp_sum_step<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// WPR = waves per row (1, 2 or 4).  A 256-thread block handles 4 / WPR rows.
template <bool RMS, int WPR, int MAXCH>
__global__ __launch_bounds__(NORM_THREADS) void norm_bf16_kernel(const uint16_t *__restrict__ x, int ldx,
                                                                 const uint16_t *__restrict__ w,
                                                                 const uint16_t *__restrict__ b, uint16_t *__restrict__ y,
                                                                 int ldy, long rows, int C, float eps,
                                                                 const uint16_t *__restrict__ w2, int G, NormGather ps)
{
    // G column groups of C elements per memory row (G = 2: the q and k blocks of a qkv row, weights w / w2): the kernel's
    // "rows" are (memory row, group) pairs
    __shared__ float red[4][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows_per_block = 4 / WPR;
    const int rloc = wave / WPR, wsub = wave % WPR;
    const long row = (long)blockIdx.x * rows_per_block + rloc;
    const bool live = row < rows;
    const long rv = live ? row : rows - 1;
    const long r = G == 1 ? rv : rv / G;
    const int grp = G == 1 ? 0 : (int)(rv - r * G);
    if (grp) w = w2;
    const int nchunk = C >> 3;
    const uint16_t *xr = x + r * (long)ldx + (long)grp * C;
    // pixel-shuffled rows (ps.hw > 0; G == 1): chunk c of the row lives in token (2 i2 + a) * hw + 2 j2 + b of tile n, quad = c / cseg
    const int ps_h2 = ps.hw >> 1;
    const long ps_n = ps.hw ? r / ((long)ps_h2 * ps_h2) : 0;
    const int ps_ij = ps.hw ? (int)(r - ps_n * ps_h2 * ps_h2) : 0;
    const int ps_i2 = ps.hw ? ps_ij / ps_h2 : 0, ps_j2 = ps.hw ? ps_ij - ps_i2 * ps_h2 : 0;
    auto chunk_src = [&](int c) -> const uint16_t * {
        if (!ps.hw) return xr + c * 8;
        const int quad = (c >= ps.cseg) + (c >= 2 * ps.cseg) + (c >= 3 * ps.cseg);
        const long tok = ps.tok0 + (long)(2 * ps_i2 + (quad >> 1)) * ps.hw + (2 * ps_j2 + (quad & 1));
        return x + ps_n * ps.tile_stride + tok * (long)ldx + (c - quad * ps.cseg) * 8;
    };

    // (weight / bias are requested behind the statistics, in the store loop: asking for them together with the row -- they do not
    //  depend on the statistics -- measured 22.8 instead of 21.5 us per launch inside the ViT-L step, same box, round 3)
    uint4_t v[MAXCH];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = (i * WPR + wsub) * 64 + lane;
        if (c < nchunk) {
            v[i] = *reinterpret_cast<const uint4_t *>(chunk_src(c));
            const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = bf16lo_to_f32(u[k]), bb = bf16hi_to_f32(u[k]);
                s += a + bb;
                ss += a * a + bb * bb;
            }
        }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (WPR > 1) {
        if (lane == 0) { red[wave][0] = s; red[wave][1] = ss; }
        __syncthreads();
        s = 0.f; ss = 0.f;
#pragma unroll
        for (int k = 0; k < WPR; ++k) { s += red[rloc * WPR + k][0]; ss += red[rloc * WPR + k][1]; }
    }
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(ss / (float)C + eps);
    } else {
        mean = s / (float)C;
        // second moment about the mean from the cached registers (no cancellation): one more cheap pass
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = (i * WPR + wsub) * 64 + lane;
            if (c < nchunk) {
                const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf16lo_to_f32(u[k]) - mean, bb = bf16hi_to_f32(u[k]) - mean;
                    var += a * a + bb * bb;
                }
            }
        }
        var = wave_sum(var);
        if (WPR > 1) {
            __syncthreads();
            if (lane == 0) red[wave][0] = var;
            __syncthreads();
            var = 0.f;
#pragma unroll
            for (int k = 0; k < WPR; ++k) var += red[rloc * WPR + k][0];
        }
        rstd = rsqrtf(var / (float)C + eps);
    }
    if (!live) return;
    uint16_t *yr = y + r * (long)ldy + (long)grp * C;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = (i * WPR + wsub) * 64 + lane;
        if (c < nchunk) {
            const uint4_t wv = *reinterpret_cast<const uint4_t *>(w + c * 8);
            uint4_t bv = {0, 0, 0, 0};
            if (!RMS && b) bv = *reinterpret_cast<const uint4_t *>(b + c * 8);
            const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            const uint32_t uw[4] = {wv.x, wv.y, wv.z, wv.w};
            const uint32_t ub[4] = {bv.x, bv.y, bv.z, bv.w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = bf16lo_to_f32(u[k]), bb = bf16hi_to_f32(u[k]);
                if (RMS) {
                    // reference order: normalise in fp32, cast to bf16, THEN multiply by the bf16 weight
                    a = bf16_to_f32(f32_to_bf16(a * rstd)) * bf16lo_to_f32(uw[k]);
                    bb = bf16_to_f32(f32_to_bf16(bb * rstd)) * bf16hi_to_f32(uw[k]);
                } else {
                    a = (a - mean) * rstd * bf16lo_to_f32(uw[k]) + bf16lo_to_f32(ub[k]);
                    bb = (bb - mean) * rstd * bf16hi_to_f32(uw[k]) + bf16hi_to_f32(ub[k]);
                }
                o[k] = pack_bf16x2(a, bb);
            }
            uint4_t ov; ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
            *reinterpret_cast<uint4_t *>(yr + c * 8) = ov;
        }
    }
}

int norm_bf16_launch(bool rms, const uint16_t *x, int ldx, const uint16_t *w, const uint16_t *b, uint16_t *y, int ldy,
                     long rows, int C, float eps, hipStream_t st, const uint16_t *w2, int G, const NormGather *psp)
{
    if (rows == 0) return VLLM_OK;
    NormGather ps = {0, 0, 0, 0};
    if (psp) {
        ps = *psp;
        VLLM_REQUIRE(G == 1 && ps.hw > 0 && ps.hw % 2 == 0 && ps.cseg > 0 && ps.cseg * 4 * 8 == C && rows % ((long)(ps.hw / 2) * (ps.hw / 2)) == 0,
                     "norm: pixel-shuffled rows need an even token grid, C = 4 segments, whole tiles");
    }
    VLLM_REQUIRE(x && w && y, "norm: null pointer");
    VLLM_REQUIRE(G == 1 || (G == 2 && w2 && aligned16(w2) && !b), "norm: column groups: G = 2 with a second weight and no bias");
    rows *= G;
    VLLM_REQUIRE(C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(w) &&
                     (!b || aligned16(b)),
                 "norm: C and row strides must be multiples of 8 elements and pointers 16-byte aligned (C=%d)", C);
    const int nchunk = C / 8;
    int wpr = 1;
    while (wpr < 4 && nchunk > 64 * NORM_MAX_CHUNKS * wpr) wpr <<= 1;
    VLLM_REQUIRE(nchunk <= 64 * NORM_MAX_CHUNKS * wpr, "norm: C=%d too wide (max %d)", C, 64 * NORM_MAX_CHUNKS * 4 * 8);
    // prefer <= 4 cached chunks per lane (58 VGPRs, full occupancy) by spreading wide rows over 2 or 4 waves
    while (wpr < 4 && nchunk > 64 * 4 * wpr) wpr <<= 1;
    const int per_lane = (nchunk + 64 * wpr - 1) / (64 * wpr);
    const int maxch = per_lane <= 1 ? 1 : per_lane <= 2 ? 2 : per_lane <= 4 ? 4 : 8;
    const int rpb = 4 / wpr;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(NORM_THREADS);
#define L3(R, W, M) VLLM_LAUNCH((norm_bf16_kernel<R, W, M>), grid, block, 0, st, x, ldx, w, b, y, ldy, rows, C, eps, w2, G, ps)
#define L2(R, W) do { if (maxch == 1) L3(R, W, 1); else if (maxch == 2) L3(R, W, 2); else if (maxch == 4) L3(R, W, 4); else L3(R, W, 8); } while (0)
    if (rms) { if (wpr == 1) L2(true, 1); else if (wpr == 2) L2(true, 2); else L2(true, 4); }
    else     { if (wpr == 1) L2(false, 1); else if (wpr == 2) L2(false, 2); else L2(false, 4); }
#undef L2
#undef L3
    VLLM_CHECK_LAUNCH("norm_bf16_kernel");
    return VLLM_OK;
}

}  // namespace vllm

using namespace vllm;

extern "C" int vllm_rmsnorm_bf16(const uint16_t *x, int ldx, const uint16_t *weight, uint16_t *y, int ldy, long rows,
                                 int C, float eps, vllm_stream_t stream)
{
    return norm_bf16_launch(true, x, ldx, weight, nullptr, y, ldy, rows, C, eps, (hipStream_t)stream);
}

extern "C" int vllm_layernorm_bf16(const uint16_t *x, int ldx, const uint16_t *weight, const uint16_t *bias,
                                   uint16_t *y, int ldy, long rows, int C, float eps, vllm_stream_t stream)
{
    return norm_bf16_launch(false, x, ldx, weight, bias, y, ldy, rows, C, eps, (hipStream_t)stream);
}
