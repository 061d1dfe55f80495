// The deformable-attention LAYER around the MSDA operator for bf16 modules:
//   value_proj (+ key-padding zero fill) -> sampling_offsets / attention_weights linears -> softmax over L*P ->
//   location arithmetic -> operator -> output_proj.
// Reference: MSDeformAttn.forward  unipose/ops/modules/ms_deform_attn.py:83-145;  mmcv MultiScaleDeformableAttention.forward
// mmcv/ops/multi_scale_deform_attn.py:262-367;  GroundingDinoMultiscaleDeformableAttention.forward ...mask_dn.py:706-784.
//
// The four linears run on the bf16 MFMA GEMM with the fp32 epilogue (VLLM_EPI_F32): value, offsets and logits never
// pass through bf16, so the sampling geometry is as accurate as the reference's fp32 upcast around the operator.
// Two elementwise kernels sit between the GEMMs and the operator:
//   msda_prep_kernel   in place: offsets -> locations, logits -> softmax weights     (HBM-bound, one read + one write)
//   f32_to_bf16_kernel operator output -> the bf16 operand of output_proj
// Both disappear in the common case (L * P == 16, P even, H * L * P * 2 a multiple of 128 -- every detection head of the
// reference): sampling_offsets and attention_weights become ONE GEMM over the queries whose epilogue (EPI_MSDA,
// gemm_epilogue.hpp) does the softmax and the location arithmetic on the accumulator, and the LDS-tiled operator writes
// bf16 itself.  The layer is then 4 launches: value GEMM, query GEMM, operator, output GEMM.
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

int msda_forward_f32_out16(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                           int B, int S, int M, int D, int L, int Lq, int P, float *out, uint16_t *out16, int *where,
                           hipStream_t st, int geometry);   // msda.hip

namespace {

// One thread per (row, head, level): the P points of that level (P * 2 offsets, P logits) are contiguous, the L threads
// of a (row, head) sit in consecutive lanes and exchange the softmax statistics with shuffles.  L is a power of two <= 8.
template <int P, int L>
__global__ __launch_bounds__(256) void msda_prep_kernel(float *__restrict__ off, float *__restrict__ lg,
                                                        const float *__restrict__ ref, const int64_t *__restrict__ shapes,
                                                        long RM, int M, int ref_dim, int four_d)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;   // ((row * M + m) * L + l)
    const bool live = t < RM * L;
    const long tt = live ? t : RM * L - 1;
    const int l = (int)(tt % L);
    const long rm = tt / L, row = rm / M;
    const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
    const float *r = ref + (row * L + l) * ref_dim;
    const float rx = r[0], ry = r[1];
    float sx, sy;   // location = ref + offset * (sx, sy)
    if (ref_dim == 2) { sx = 1.f / W; sy = 1.f / H; }
    else if (four_d) { sx = r[2] * 0.5f / W; sy = r[3] * 0.5f / H; }
    else { sx = r[2] * 0.5f / (float)P; sy = r[3] * 0.5f / (float)P; }
    float o[P * 2], g[P];
#pragma unroll
    for (int i = 0; i < P * 2; ++i) o[i] = off[tt * (P * 2) + i];
#pragma unroll
    for (int i = 0; i < P; ++i) g[i] = lg[tt * P + i];
    float mx = g[0];
#pragma unroll
    for (int i = 1; i < P; ++i) mx = fmaxf(mx, g[i]);
#pragma unroll
    for (int d = 1; d < L; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) { g[i] = __expf(g[i] - mx); sum += g[i]; }
#pragma unroll
    for (int d = 1; d < L; d <<= 1) sum += __shfl_xor(sum, d);
    const float inv = 1.f / sum;
    if (!live) return;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        off[tt * (P * 2) + 2 * i] = rx + o[2 * i] * sx;
        off[tt * (P * 2) + 2 * i + 1] = ry + o[2 * i + 1] * sy;
        lg[tt * P + i] = g[i] * inv;
    }
}

// Any L, P: one thread per (row, head).
__global__ __launch_bounds__(256) void msda_prep_generic_kernel(float *__restrict__ off, float *__restrict__ lg,
                                                                const float *__restrict__ ref,
                                                                const int64_t *__restrict__ shapes, long RM, int M, int L,
                                                                int P, int ref_dim, int four_d)
{
    const long rm = (long)blockIdx.x * 256 + threadIdx.x;
    if (rm >= RM) return;
    const long row = rm / M;
    float *g = lg + rm * L * P, *o = off + rm * L * P * 2;
    float mx = g[0];
    for (int i = 1; i < L * P; ++i) mx = fmaxf(mx, g[i]);
    float sum = 0.f;
    for (int i = 0; i < L * P; ++i) sum += __expf(g[i] - mx);
    const float inv = 1.f / sum;
    for (int l = 0; l < L; ++l) {
        const float W = (float)shapes[2 * l + 1], H = (float)shapes[2 * l];
        const float *r = ref + (row * L + l) * ref_dim;
        float sx, sy;
        if (ref_dim == 2) { sx = 1.f / W; sy = 1.f / H; }
        else if (four_d) { sx = r[2] * 0.5f / W; sy = r[3] * 0.5f / H; }
        else { sx = r[2] * 0.5f / (float)P; sy = r[3] * 0.5f / (float)P; }
        for (int p = 0; p < P; ++p) {
            const int i = l * P + p;
            o[2 * i] = r[0] + o[2 * i] * sx;
            o[2 * i + 1] = r[1] + o[2 * i + 1] * sy;
            g[i] = __expf(g[i] - mx) * inv;
        }
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, long n4,
                                                          long n, const int64_t *__restrict__ shapes, int L, long Lq)
{
    // shapes != null: the operator already wrote bf16 for a pyramid geometry (msda_forward_f32_out16) -- nothing to do then.
    // Grid-stride over a fixed grid, so that this "nothing" costs a couple of microseconds instead of 75 k empty blocks.
    if (shapes && geometry_is_nested(shapes, L, Lq)) return;   // (the predicate of the pyramid-item kernels, which then wrote bf16 themselves)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4_t v = reinterpret_cast<const float4_t *>(src)[i];
        uint2_t o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        reinterpret_cast<uint2_t *>(dst)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long k = n4 * 4; k < n; ++k) dst[k] = f32_to_bf16(src[k]);
}

// msda_layer_fused = 0 (option / VLLM_MSDA_LAYER_FUSED): the round-1 composition (two query GEMMs + prep kernel + fp32
// operator + conversion pass), kept as the A/B reference of the fused epilogue.
inline bool layer_unfused() { return msda_layer_fused() == 0; }
// VLLM_MSDA_LAYER_VALUE_BF16=0: keep the value in fp32 for every query set (A/B)
static int g_layer_value_bf16 = -1;
inline bool msda_layer_value_bf16()
{
    if (g_layer_value_bf16 < 0) { const char *e = getenv("VLLM_MSDA_LAYER_VALUE_BF16"); g_layer_value_bf16 = e && e[0] == '0' ? 0 : 1; }
    return g_layer_value_bf16 != 0;
}

inline long align256(long x) { return (x + 255) & ~255L; }

struct LayerWs {
    long value, off, logit, opout, opout_bf16, total;
};
LayerWs layer_ws(const VllmMsdaLayerDesc *d, long B, long Lq, long S)
{
    const long C = d->d_model, MLP = (long)d->n_heads * d->n_levels * d->n_points;
    LayerWs w;
    long p = 0;
    w.value = p; p += align256(B * S * C * 4);
    w.off = p; p += align256(B * Lq * MLP * 2 * 4);
    w.logit = p; p += align256(B * Lq * MLP * 4);
    w.opout = p; p += align256(B * Lq * C * 4);
    w.opout_bf16 = p; p += align256(B * Lq * C * 2);
    w.total = p;
    return w;
}

int prep_launch(float *off, float *lg, const float *ref, const int64_t *shapes, long R, int M, int L, int P, int ref_dim,
                int four_d, hipStream_t st)
{
    const long RM = R * M;
    if (RM == 0) return VLLM_OK;
    VLLM_REQUIRE(RM * L < (1L << 38), "msda_prep: too many rows");
#define GO(PP, LL)                                                                                       \
    VLLM_LAUNCH((msda_prep_kernel<PP, LL>), dim3((unsigned)ceil_div(RM * LL, 256)), dim3(256), 0, st, off, lg, ref, shapes, \
                RM, M, ref_dim, four_d)
    if (P == 4 && L == 4) GO(4, 4);
    else if (P == 4 && L == 1) GO(4, 1);
    else if (P == 4 && L == 2) GO(4, 2);
    else if (P == 4 && L == 8) GO(4, 8);
    else if (P == 8 && L == 4) GO(8, 4);
    else
        VLLM_LAUNCH(msda_prep_generic_kernel, dim3((unsigned)ceil_div(RM, 256)), dim3(256), 0, st, off, lg, ref, shapes, RM,
                    M, L, P, ref_dim, four_d);
#undef GO
    VLLM_CHECK_LAUNCH("msda_prep_kernel");
    return VLLM_OK;
}

int cvt_launch(const float *src, uint16_t *dst, long n, hipStream_t st, const int64_t *shapes = nullptr, int L = 0, long Lq = 0)
{
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "f32_to_bf16: unaligned operands");
    const long n4 = n / 4;
    const long blocks = ceil_div(n4 > 0 ? n4 : 1, 256);
    VLLM_LAUNCH(f32_to_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, src, dst, n4, n, shapes, L, Lq);
    VLLM_CHECK_LAUNCH("f32_to_bf16_kernel");
    return VLLM_OK;
}

}  // namespace
int msda_layer_value_bf16_set(int v) { const int old = msda_layer_value_bf16() ? 1 : 0; g_layer_value_bf16 = v != 0; return old; }
}  // namespace vllm

using namespace vllm;

#define TRY(x)                      \
    do {                            \
        const int rc__ = (x);       \
        if (rc__ != VLLM_OK) return rc__; \
    } while (0)

extern "C" unsigned long vllm_msda_layer_desc_sizeof(void) { return sizeof(VllmMsdaLayerDesc); }

static int check_desc(const VllmMsdaLayerDesc *d)
{
    VLLM_REQUIRE(d, "msda_layer: null descriptor");
    VLLM_REQUIRE(d->d_model > 0 && d->n_heads > 0 && d->n_levels > 0 && d->n_points > 0, "msda_layer: bad sizes");
    VLLM_REQUIRE(d->d_model % d->n_heads == 0, "embed_dims must be divisible by num_heads, but got %d and %d", d->d_model,
                 d->n_heads);   // ms_deform_attn.py:52-53, multi_scale_deform_attn.py:208-210
    VLLM_REQUIRE(d->d_model % 64 == 0, "msda_layer: d_model=%d must be a multiple of 64 (GEMM K tile)", d->d_model);
    VLLM_REQUIRE((d->d_model / d->n_heads) % 4 == 0, "msda_layer: head dimension must be a multiple of 4");
    VLLM_REQUIRE(d->ref_dim == 2 || d->ref_dim == 4, "Last dim of reference_points must be 2 or 4, but get %d instead.",
                 d->ref_dim);   // ms_deform_attn.py:127-129
    VLLM_REQUIRE(d->value_proj_w && d->sampling_offsets_w && d->attention_weights_w && d->output_proj_w,
                 "msda_layer: missing weights");
    return VLLM_OK;
}

extern "C" long vllm_msda_layer_workspace_bytes(const VllmMsdaLayerDesc *d, int B, int Lq, int S)
{
    if (check_desc(d) != VLLM_OK || B < 0 || Lq < 0 || S < 0) return -1;
    return layer_ws(d, B, Lq, S).total;
}

extern "C" int vllm_msda_prep_f32(float *off, float *lg, const float *ref, const int64_t *shapes, long R, int M, int L, int P,
                                  int ref_dim, int four_d, vllm_stream_t stream)
{
    VLLM_REQUIRE(R >= 0 && M > 0 && L > 0 && P > 0 && (ref_dim == 2 || ref_dim == 4), "msda_prep: bad sizes");
    if (R == 0) return VLLM_OK;
    VLLM_REQUIRE(off && lg && ref && shapes, "msda_prep: null pointer");
    return prep_launch(off, lg, ref, shapes, R, M, L, P, ref_dim, four_d, (hipStream_t)stream);
}

extern "C" int vllm_f32_to_bf16(const float *src, uint16_t *dst, long n, vllm_stream_t stream)
{
    VLLM_REQUIRE(n >= 0 && (n == 0 || (src && dst)), "f32_to_bf16: bad arguments");
    return cvt_launch(src, dst, n, (hipStream_t)stream);
}

extern "C" int vllm_msda_layer_forward(const VllmMsdaLayerDesc *d, const uint16_t *query, const float *ref,
                                       const uint16_t *input_flatten, const uint8_t *padding_mask, const int64_t *shapes,
                                       const int64_t *lsi, int B, int Lq, int S, uint16_t *out, void *workspace, long ws_bytes,
                                       vllm_stream_t stream)
{
    TRY(check_desc(d));
    VLLM_REQUIRE(B >= 0 && Lq >= 0 && S >= 0, "msda_layer: negative size");
    if (B == 0 || Lq == 0) return VLLM_OK;
    VLLM_REQUIRE(S > 0, "msda_layer: empty value");
    VLLM_REQUIRE(query && ref && input_flatten && shapes && lsi && out, "msda_layer: null pointer");
    const LayerWs w = layer_ws(d, B, Lq, S);
    VLLM_REQUIRE(workspace && ws_bytes >= w.total, "msda_layer: workspace too small (%ld < %ld)", ws_bytes, w.total);
    VLLM_REQUIRE((long)B * S < (1L << 31) && (long)B * Lq < (1L << 31), "msda_layer: too many rows for one launch");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int C = d->d_model, M = d->n_heads, L = d->n_levels, P = d->n_points, D = C / M, MLP = M * L * P;
    float *value = (float *)(ws + w.value), *off = (float *)(ws + w.off), *lg = (float *)(ws + w.logit);
    float *opout = (float *)(ws + w.opout);
    uint16_t *opb = (uint16_t *)(ws + w.opout_bf16);
    prof_mark(PT_MSDA_LAYER, st);
    // value = value_proj(input_flatten), padded keys zeroed (ms_deform_attn.py:106-109) -- fp32 [B, S, M, D] where the LDS-tiled
    // operator can run (its windows are fp32: a bf16 value would cost its gather conversions it has no issue slots for, NOTES/rounds_1_to_4.md 3.6).
    // A query set that is NOT the value pyramid (decoder cross-attention: Lq != S) runs the gather kernel whatever the value's
    // dtype, and that one reads bf16 natively: the value is then stored in bf16 -- what the reference's own bf16 module computes
    // (...mask_dn.py:764-766 rounds it to bf16 before the upcast) at half the value GEMM's write and half the operator's read.
    uint16_t *value16 = (uint16_t *)value;
    GemmArgs va;
    {
        va.X = input_flatten; va.W = d->value_proj_w; va.Y = value16; va.bias = d->value_proj_b; va.scale = nullptr; va.res = nullptr;
        va.M = B * S; va.N = C; va.K = C; va.ldx = C; va.ldw = C; va.ldy = C; va.ldr = 0; va.P = 0; va.mt = va.nt = 0; va.xP = 0;
        va.variant = 0; va.variant256 = 0; va.direct_store = 0; va.row_mask = padding_mask;
    }
    // (the bf16-value operator needs 8 channels per lane: D / 8 a power of two, 8 <= D <= 512 -- vllm_msda_forward_bf16's own
    //  precondition; a d_model-256 layer with 64 heads (D = 4) keeps the fp32 value path instead of failing there.  ADVICE r4)
    const bool d_bf16_ok = D >= 8 && D <= 512 && D % 8 == 0 && ((D / 8) & (D / 8 - 1)) == 0;
    const bool v16 = Lq != S && d_bf16_ok && msda_layer_value_bf16() && gemm_skinny_takes(EPI_BIAS, va);
    if (v16) TRY(gemm_bf16_launch(EPI_BIAS, va, st));
    else
    TRY(gemm(st, EPI_F32, input_flatten, C, d->value_proj_w, C, d->value_proj_b, (uint16_t *)value, C, B * S, C, C, nullptr,
             (const uint16_t *)padding_mask));
    // offsets and logits of the queries (:110-111) in one GEMM; softmax + location arithmetic (:112-129) in its epilogue: the tile
    // kernel's form needs L * P == 16; the streaming kernel's (gemm_skinny.hip: d_model 256, 8 heads, 4 points, >= 4096 query
    // rows) also takes 1 ... 3 levels -- the 3-level pixel decoder (msdeformattn_pixel_decoder.py:57-58)
    GemmArgs a;
    {
        a.X = query; a.W = d->sampling_offsets_w; a.Y = (uint16_t *)off; a.bias = d->sampling_offsets_b; a.scale = nullptr;
        a.res = nullptr; a.M = B * Lq; a.N = MLP * 3; a.K = C; a.ldx = C; a.ldw = C; a.ldy = MLP * 2; a.ldr = 0; a.P = 0;
        a.mt = a.nt = 0; a.xP = 0; a.variant = 0; a.variant256 = 0; a.direct_store = 0;
        a.W2 = d->attention_weights_w; a.bias2 = d->attention_weights_b; a.Y2 = lg; a.ldy2 = MLP; a.nsplit = MLP * 2;
        a.ref = ref; a.shapes = shapes; a.mL = L; a.mP = P; a.ref_dim = d->ref_dim; a.four_d = d->use_4d_normalizer;
    }
    const bool tile_form = L * P == 16 && P % 2 == 0 && (MLP * 2) % 128 == 0;
    if (!layer_unfused() && (tile_form || gemm_skinny_takes(EPI_MSDA, a))) {
        TRY(gemm_bf16_launch(EPI_MSDA, a, st));
    } else {
        // offsets / logits of the queries (:110-111) -- fp32 [B*Lq, M*L*P*2], [B*Lq, M*L*P]
        TRY(gemm(st, EPI_F32, query, C, d->sampling_offsets_w, C, d->sampling_offsets_b, (uint16_t *)off, MLP * 2, B * Lq,
                 MLP * 2, C));
        TRY(gemm(st, EPI_F32, query, C, d->attention_weights_w, C, d->attention_weights_b, (uint16_t *)lg, MLP, B * Lq, MLP,
                 C));
        // softmax + location arithmetic, in place (:112-129)
        TRY(prep_launch(off, lg, ref, shapes, (long)B * Lq, M, L, P, d->ref_dim, d->use_4d_normalizer, st));
    }
    // the operator (:131-139), fp32 arithmetic; bf16 result straight from the LDS-tiled kernel where that one runs
    int where = 0;
    if (v16) TRY(vllm_msda_forward_bf16(value16, shapes, lsi, off, lg, B, S, M, D, L, Lq, P, opb, stream));   // (bf16 result: output_proj's operand)
    else if (layer_unfused()) {
        TRY(vllm_msda_forward_f32(value, shapes, lsi, off, lg, B, S, M, D, L, Lq, P, opout, stream));
        prof_mark(PT_MSDA_LAYER, st);   // (the public entry point marks its own tag and closes it: the layer's tag goes on behind it)
    }
    else TRY(msda_forward_f32_out16(value, shapes, lsi, off, lg, B, S, M, D, L, Lq, P, opout, opb, &where, st, d->geometry));
    // output_proj (:144).  The conversion pass runs unless the host KNOWS the operator wrote bf16 (pyramid hint); with an
    // unknown geometry it is enqueued and tests the device-side predicate itself.
    if (!v16 && !(where && (d->geometry == VLLM_GEO_PYRAMID || d->geometry == VLLM_GEO_NESTED)))
        TRY(cvt_launch(opout, opb, (long)B * Lq * C, st, where ? shapes : nullptr, L, Lq));
    TRY(gemm(st, EPI_BIAS, opb, C, d->output_proj_w, C, d->output_proj_b, out, C, B * Lq, C, C));
    prof_mark(PT_END, st);
    return VLLM_OK;
}
