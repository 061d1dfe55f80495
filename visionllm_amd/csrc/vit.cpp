// Encoder / projector orchestration behind the C ABI: one host call enqueues every kernel of the image -> visual
// token path on the caller's stream (no allocation, no synchronisation; workspace provided by the caller).
//
// vllm_vit_forward replaces the Python layer loop of InternVisionEncoder.forward
// (VisionLLMv2/visionllmv2/model/internvit/modeling_intern_vit.py:253-270) and of HF CLIPEncoder;
// vllm_bridge_forward replaces select + pixel_shuffle + vl_bridge (visionllmv2/model/modeling_visionllmv2.py:569-579).
#include <stdlib.h>
#include <algorithm>
#include "kernels.hpp"

using namespace vllm;

namespace {
inline long align256(long x) { return (x + 255) & ~255L; }

struct VitWs {
    long xn, qkv, ao, hmid, mid, col, h0, h1, sk, ln, total;
};

VitWs vit_ws_layout(const VllmVitDesc *d, int n)
{
    const long g = d->image / d->patch, P = g * g, S = P + 1, M = (long)n * S;
    VitWs w;
    long off = 0;
    auto take = [&](long bytes) { long o = off; off += align256(bytes); return o; };
    w.xn = take(M * d->hidden * 2);
    w.qkv = take(M * 3L * d->hidden * 2);
    w.ao = take(M * d->hidden * 2);
    w.hmid = take(M * d->hidden * 2);
    w.mid = take(M * (long)d->inter * 2);
    w.col = take((long)n * P * d->kpad * 2);
    w.h0 = take(M * d->hidden * 2);   // ping-pong hidden states for entries the caller does not want
    w.h1 = take(M * d->hidden * 2);
    w.sk = take(SK_SCRATCH_BYTES);    // stream-K tail of the GEMMs (kernels.hpp)
    w.ln = take(2 * M * std::max((d->hidden + 255) / 256 * 2, 16) * 4);   // folded norms: {mean, M2} per (row, 256-column tile) or the wide layout's 16 floats per row; two buffers
    w.total = off;
    return w;
}

int check_desc(const VllmVitDesc *d)
{
    VLLM_REQUIRE(d, "vit: null descriptor");
    VLLM_REQUIRE(d->arch == VLLM_ARCH_INTERNVIT || d->arch == VLLM_ARCH_CLIP, "vit: unknown arch %d", d->arch);
    VLLM_REQUIRE(d->hidden > 0 && d->heads > 0 && d->hidden % d->heads == 0, "vit: hidden %d / heads %d", d->hidden, d->heads);
    const int hd = d->hidden / d->heads;
    VLLM_REQUIRE(hd == 64 || hd == 128, "vit: head_dim %d not supported by the attention kernel (64 or 128)", hd);
    VLLM_REQUIRE(d->hidden % 64 == 0 && d->inter % 64 == 0, "vit: hidden/intermediate must be multiples of 64");
    VLLM_REQUIRE(d->patch > 0 && d->image % d->patch == 0, "vit: image %d not divisible by patch %d", d->image, d->patch);
    VLLM_REQUIRE(d->kpad % 64 == 0 && d->kpad >= 3 * d->patch * d->patch, "vit: kpad %d", d->kpad);
    VLLM_REQUIRE(d->num_layers >= 0 && (d->num_layers == 0 || d->layers), "vit: layers missing");
    VLLM_REQUIRE(d->patch_w && d->cls && d->pos, "vit: embedding parameters missing");
    VLLM_REQUIRE(d->act == VLLM_EPI_GELU || d->act == VLLM_EPI_QUICK_GELU, "vit: act %d", d->act);
    return VLLM_OK;
}
}  // namespace

extern "C" int vllm_vit_desc_sizeof(void) { return (int)sizeof(VllmVitDesc); }
extern "C" int vllm_vit_layer_sizeof(void) { return (int)sizeof(VllmVitLayer); }
extern "C" int vllm_bridge_desc_sizeof(void) { return (int)sizeof(VllmBridgeDesc); }

extern "C" long vllm_vit_workspace_bytes(const VllmVitDesc *d, int n_tiles)
{
    if (check_desc(d) != VLLM_OK || n_tiles < 0) return -1;
    return vit_ws_layout(d, n_tiles).total;
}

#define TRY(x) do { int rc__ = (x); if (rc__ != VLLM_OK) return rc__; } while (0)

static long g_folded_gemms = 0;   // GEMM launches with a norm folded in (tests assert the path they mean to cover ran)
extern "C" long vllm_vit_folded_gemm_launches(void) { return __atomic_load_n(&g_folded_gemms, __ATOMIC_RELAXED); }

extern "C" int vllm_vit_forward(const VllmVitDesc *d, const void *pixels, int n, uint16_t *const *hs, void *workspace,
                                long ws_bytes, vllm_stream_t stream)
{
    TRY(check_desc(d));
    VLLM_REQUIRE(n >= 0, "vit: negative tile count");
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(pixels && hs && workspace, "vit: null pointer");
    const VitWs w = vit_ws_layout(d, n);
    VLLM_REQUIRE(ws_bytes >= w.total, "vit: workspace too small (%ld < %ld)", ws_bytes, w.total);
    VLLM_REQUIRE(hs[d->num_layers], "vit: the last hidden state must be provided");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int C = d->hidden, H = d->heads, D = C / H, I = d->inter;
    const int g = d->image / d->patch, P = g * g, S = P + 1;
    const long M = (long)n * S;
    VLLM_REQUIRE(M < (1L << 31) / 4, "vit: too many tokens");
    uint16_t *xn = (uint16_t *)(ws + w.xn), *qkv = (uint16_t *)(ws + w.qkv), *ao = (uint16_t *)(ws + w.ao);
    uint16_t *hmid = (uint16_t *)(ws + w.hmid), *mid = (uint16_t *)(ws + w.mid), *col = (uint16_t *)(ws + w.col);
    uint16_t *pp[2] = {(uint16_t *)(ws + w.h0), (uint16_t *)(ws + w.h1)};
    const bool clip = d->arch == VLLM_ARCH_CLIP;
    const float scale = 1.0f / sqrtf((float)D);

    auto state = [&](int i) -> uint16_t * { return hs[i] ? hs[i] : pp[i & 1]; };

    void *sk = ws + w.sk;
    VLLM_REQUIRE(hipMemsetAsync(sk, 0, SK_FLAG_BYTES, st) == hipSuccess, "vit: flag reset failed");

    // ---- embeddings: im2col gather -> GEMM (+bias +pos, rows scattered past CLS) ; CLS rows ----
    uint16_t *emb = clip ? hmid : state(0);   // CLIP: pre_layrnorm produces hidden_states[0]
    prof_mark(PT_EMBED, st);
    TRY(im2col_launch(pixels, d->pixel_is_f32, col, n, d->image, d->patch, d->kpad, st));
    TRY(gemm(st, EPI_EMBED, col, d->kpad, d->patch_w, d->kpad, d->patch_b, emb, C, n * P, C, d->kpad, nullptr, d->pos, C, P, 0, sk, SK_SCRATCH_BYTES));
    TRY(cls_rows_launch(d->cls, d->pos, emb, n, S, C, st));
    if (clip) {
        VLLM_REQUIRE(d->pre_ln_w && d->pre_ln_b, "vit: CLIP needs pre_layrnorm");
        prof_mark(PT_NORM, st);
        TRY(norm_bf16_launch(false, emb, C, d->pre_ln_w, d->pre_ln_b, state(0), C, M, C, d->eps, st));
    }

    // Folded norms (round 3): between a residual GEMM and the qkv / fc1 GEMM behind it the norm is not launched -- the residual
    // GEMM's epilogue leaves per-row statistics of what it stores (ln_a: for norm1 of the next layer, ln_b: for norm2 of this one),
    // the consumer multiplies the un-normalised rows with gamma-scaled weights and applies the statistics in its epilogue
    // (GemmArgs::ln_*; 46 of the 49 norm launches of a ViT-L forward pass disappear).  Needs the prepared weights in the
    // descriptor and shapes that take the 8-phase GEMM; VLLM_LN_FOLD=0 switches it off (A/B).
    static const int fold_off = [] { const char *e = getenv("VLLM_LN_FOLD"); return e && e[0] == '0' ? 1 : 0; }();
    const int ntC = (C + 255) / 256;
    // (two buffers; the wide layout is 16 floats per row whatever ntC -- M * 16 <= M * ntC * 2 floats for ntC >= 8 -- and keeps ln_b 16-byte aligned for odd M)
    float *ln_a = (float *)(ws + w.ln), *ln_b = ln_a + (size_t)M * ((d->arch != VLLM_ARCH_CLIP && C != 1024) ? 16 : ntC * 2);
    // (the consumer stages the statistics of rows of exactly four 256-column tiles: hidden size 1024 -- ViT-L, InternViT-300M.
    //  Round 5: RMSNorm rows of up to 16 column tiles -- InternViT-6B, hidden 3200 = 12.5 tiles -- in the WIDE format, one float per
    //  (row, tile), which only the persistent GEMM schedule implements: the fold is on when all four GEMMs of a layer take that
    //  schedule at this batch size (asked below with a dry run), otherwise the norms are launched as before.)
    const bool wide = !clip && C != 1024;
    bool shapes_fold = !fold_off && M >= 1024 && I >= 1024 && I % 8 == 0 && (C == 1024 || (wide && C % 8 == 0 && ntC <= 16));
    int dry = 0;
    auto gemm_ln = [&](int epi, const uint16_t *X, int ldx, const uint16_t *W, int ldw, const uint16_t *bias, uint16_t *Y, int ldy, int N,
                       int K, const uint16_t *scale, const uint16_t *res, int ldr, float *ln_out, const float *ln_in,
                       const float *colsum, const float *bias_ln) {
        GemmArgs a;
        gemm_set_scratch(a, sk, SK_SCRATCH_BYTES);
        a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.scale = scale; a.res = res;
        a.M = (int)M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.P = 0; a.mt = a.nt = 0; a.xP = 0;
        a.variant = gemm_variant_override(); a.variant256 = 0; a.direct_store = gemm_direct_store();
        if (a.variant == 1 || a.variant == 4) a.variant = 0;   // (a forced 128x128 kernel / the 32x32x16 variant cannot fold)
        a.ln_out = ln_out; a.ln_in = ln_in; a.ln_slots = ntC; a.ln_cols = C; a.ln_rms = clip ? 0 : 1; a.ln_eps = d->eps;
        a.ln_colsum = colsum; a.ln_bias = bias_ln; a.ln_wide = wide ? 1 : 0; a.dry_run = dry;
        if (!dry) __atomic_fetch_add(&g_folded_gemms, 1L, __ATOMIC_RELAXED);
        return gemm_bf16_launch(epi, a, st);
    };
    if (shapes_fold && wide && d->num_layers > 0) {
        // dry run of the four folded GEMMs of a layer with layer 0's operands (shapes and alignments are the same in every layer)
        const VllmVitLayer &L0 = d->layers[0];
        if (L0.qkv_w_ln && L0.fc1_w_ln && L0.proj_w && L0.fc2_w) {
            dry = 1;
            const bool ok = gemm_ln(EPI_BIAS, hmid, C, L0.qkv_w_ln, C, nullptr, qkv, 3 * C, 3 * C, C, nullptr, nullptr, 0, nullptr, ln_a, nullptr, L0.qkv_bias_ln) == VLLM_OK &&
                            gemm_ln(EPI_RESIDUAL, ao, C, L0.proj_w, C, L0.proj_b, hmid, C, C, C, L0.ls1, hmid, C, ln_b, nullptr, nullptr, nullptr) == VLLM_OK &&
                            gemm_ln(d->act, hmid, C, L0.fc1_w_ln, C, nullptr, mid, I, I, C, nullptr, nullptr, 0, nullptr, ln_b, nullptr, L0.fc1_bias_ln) == VLLM_OK &&
                            gemm_ln(EPI_RESIDUAL, mid, I, L0.fc2_w, I, L0.fc2_b, hmid, C, C, I, L0.ls2, hmid, C, ln_a, nullptr, nullptr, nullptr) == VLLM_OK;
            dry = 0;
            //  the dry run sees layer 0's operands and the workspace buffers; the real launches use state(i) -- which may be
            // caller-provided hidden_states buffers -- and every layer's own vectors.  The wide fold has no tile-wise fall-back, so a
            // single operand the persistent schedule would refuse (16-byte alignment) switches the fold off for the whole pass.
            bool aligned = true;
            for (int i = 0; i <= d->num_layers && aligned; ++i) aligned = aligned16(state(i));
            for (int i = 0; i < d->num_layers && aligned; ++i) {
                const VllmVitLayer &Li = d->layers[i];
                const void *ops[] = {Li.qkv_w_ln, Li.fc1_w_ln, Li.proj_w, Li.fc2_w, Li.proj_b, Li.fc2_b, Li.ls1, Li.ls2, Li.qkv_bias_ln, Li.fc1_bias_ln};
                for (const void *q : ops) aligned = aligned && (!q || aligned16(q));
            }
            if (getenv("VLLM_VIT_DEBUG")) fprintf(stderr, "vit: wide norm fold dry run at M=%ld C=%d I=%d: %s (%s)\n", (long)M, C, I, ok ? "taken" : "refused", vllm_last_error());
            if (!ok || !aligned) shapes_fold = false;
        } else shapes_fold = false;
        if (shapes_fold && hipMemsetAsync(ln_a, 0, (size_t)2 * M * 16 * sizeof(float), st) != hipSuccess) {   // (the wide layout's unused slots are read as zeros)
            set_error("vit: hipMemsetAsync of the folded-norm statistics failed");
            return VLLM_ELAUNCH;
        }
    }
    auto folds = [&](int i) {   // layer i has the prepared weights
        return shapes_fold && i >= 0 && i < d->num_layers && d->layers[i].qkv_w_ln && d->layers[i].fc1_w_ln &&
               (!clip || (d->layers[i].qkv_colsum && d->layers[i].fc1_colsum));
    };

    for (int i = 0; i < d->num_layers; ++i) {
        const VllmVitLayer &L = d->layers[i];
        const uint16_t *h = state(i);
        uint16_t *hout = state(i + 1);
        VLLM_REQUIRE(L.norm1_w && L.qkv_w && L.proj_w && L.norm2_w && L.fc1_w && L.fc2_w, "vit: layer %d parameters missing", i);
        const bool fold = folds(i);
        const bool stats1 = fold && i > 0 && folds(i - 1);   // the previous layer's fc2 left norm1's statistics in ln_a
        // attention block
        if (stats1) {
            prof_mark(PT_QKV, st);
            TRY(gemm_ln(EPI_BIAS, h, C, L.qkv_w_ln, C, nullptr, qkv, 3 * C, 3 * C, C, nullptr, nullptr, 0, nullptr, ln_a, L.qkv_colsum, L.qkv_bias_ln));
        } else {
            prof_mark(PT_NORM, st);
            TRY(norm_bf16_launch(!clip, h, C, L.norm1_w, L.norm1_b, xn, C, M, C, d->eps, st));
            prof_mark(PT_QKV, st);
            TRY(gemm(st, EPI_BIAS, xn, C, L.qkv_w, C, L.qkv_b, qkv, 3 * C, (int)M, 3 * C, C, nullptr, nullptr, 0, 0, 0, sk, SK_SCRATCH_BYTES));
        }
        if (L.q_norm_w || L.k_norm_w) prof_mark(PT_QKNORM, st);
        if (L.q_norm_w && L.k_norm_w) {   // both (InternViT-6B): one launch over the [M, 2C] slab, q / k weight per column group
            TRY(norm_bf16_launch(true, qkv, 3 * C, L.q_norm_w, nullptr, qkv, 3 * C, M, C, d->eps, st, L.k_norm_w, 2));
        } else {
            if (L.q_norm_w) TRY(norm_bf16_launch(true, qkv, 3 * C, L.q_norm_w, nullptr, qkv, 3 * C, M, C, d->eps, st));
            if (L.k_norm_w) TRY(norm_bf16_launch(true, qkv + C, 3 * C, L.k_norm_w, nullptr, qkv + C, 3 * C, M, C, d->eps, st));
        }
        {
            AttnArgs a;
            a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C; a.out = ao;
            a.q_bs = a.k_bs = a.v_bs = (long)S * 3 * C;
            a.q_ts = a.k_ts = a.v_ts = 3 * C;
            a.q_hs = a.k_hs = a.v_hs = D;
            a.B = n; a.S = S; a.H = H; a.nqt = 0;
            a.scale_log2e = scale * 1.4426950408889634f;
            prof_mark(PT_ATTN, st);
            TRY(attn_fwd_launch(a, D, st));
        }
        prof_mark(PT_PROJ, st);
        if (fold) TRY(gemm_ln(EPI_RESIDUAL, ao, C, L.proj_w, C, L.proj_b, hmid, C, C, C, L.ls1, h, C, ln_b, nullptr, nullptr, nullptr));
        else TRY(gemm(st, EPI_RESIDUAL, ao, C, L.proj_w, C, L.proj_b, hmid, C, (int)M, C, C, L.ls1, h, C, 0, 0, sk, SK_SCRATCH_BYTES));
        // MLP block
        if (fold) {
            prof_mark(PT_FC1, st);
            TRY(gemm_ln(d->act, hmid, C, L.fc1_w_ln, C, nullptr, mid, I, I, C, nullptr, nullptr, 0, nullptr, ln_b, L.fc1_colsum, L.fc1_bias_ln));
        } else {
            prof_mark(PT_NORM, st);
            TRY(norm_bf16_launch(!clip, hmid, C, L.norm2_w, L.norm2_b, xn, C, M, C, d->eps, st));
            prof_mark(PT_FC1, st);
            TRY(gemm(st, d->act, xn, C, L.fc1_w, C, L.fc1_b, mid, I, (int)M, I, C, nullptr, nullptr, 0, 0, 0, sk, SK_SCRATCH_BYTES));
        }
        prof_mark(PT_FC2, st);
        if (fold && folds(i + 1)) TRY(gemm_ln(EPI_RESIDUAL, mid, I, L.fc2_w, I, L.fc2_b, hout, C, C, I, L.ls2, hmid, C, ln_a, nullptr, nullptr, nullptr));
        else TRY(gemm(st, EPI_RESIDUAL, mid, I, L.fc2_w, I, L.fc2_b, hout, C, (int)M, C, I, L.ls2, hmid, C, 0, 0, sk, SK_SCRATCH_BYTES));
    }
    prof_mark(PT_END, st);
    return VLLM_OK;
}

// ---------------------------------------------------------------------------------------------------------
namespace {
struct BridgeWs { long a, b, c, sk, sk_bytes, total; };
BridgeWs bridge_ws_layout(const VllmBridgeDesc *d, int n, int T_in)
{
    const long T = d->pixel_shuffle ? T_in / 4 : T_in;
    const long rows = (long)n * T;
    BridgeWs w;
    long off = 0;
    auto take = [&](long bytes) { long o = off; off += align256(bytes); return o; };
    w.a = take(d->pixel_shuffle ? rows * d->in_features * 2 : 0);                       // shuffled features
    w.b = take(d->kind == VLLM_BRIDGE_INTERNVL_MLP ? rows * d->in_features * 2 : 0);   // LayerNorm output
    w.c = take(d->depth > 1 ? 2 * align256(rows * (long)d->out_features * 2) : 0);     // MLP intermediates
    w.sk_bytes = rows >= 1024 ? SK_SCRATCH_BYTES : 0;                                   // stream-K tail of the 8-phase GEMM (kernels.hpp)
    w.sk = take(w.sk_bytes);
    w.total = off;
    return w;
}
}  // namespace

extern "C" long vllm_bridge_workspace_bytes(const VllmBridgeDesc *d, int n_tiles, int T_in)
{
    if (!d || n_tiles < 0 || T_in < 0) return -1;
    return bridge_ws_layout(d, n_tiles, T_in).total;
}

extern "C" int vllm_bridge_forward(const VllmBridgeDesc *d, const uint16_t *hidden, int n, int T_in, int C, uint16_t *out,
                                   void *workspace, long ws_bytes, vllm_stream_t stream)
{
    VLLM_REQUIRE(d && d->depth >= 1 && d->depth <= 4, "bridge: bad descriptor");
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(hidden && out, "bridge: null pointer");
    const BridgeWs w = bridge_ws_layout(d, n, T_in);
    VLLM_REQUIRE(w.total == 0 || (workspace && ws_bytes >= w.total), "bridge: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const int S = T_in + (d->skip_cls ? 1 : 0);
    const int Cin = d->in_features, Cout = d->out_features;
    VLLM_REQUIRE(Cin == (d->pixel_shuffle ? 4 * C : C), "bridge: in_features %d does not match C=%d", Cin, C);
    VLLM_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "bridge: feature sizes must be multiples of 64");
    const uint16_t *x = hidden;
    int ldx = C, xP = d->skip_cls ? T_in : 0;   // hidden[:, 1:] is read in place (the GEMM loader skips CLS rows)
    long rows = (long)n * T_in;
    // Round 5: with the InternVL projector behind it (LayerNorm first) the pixel-shuffle is not launched: the LayerNorm gathers the
    // 2 x 2 token neighbourhoods itself (NormGather; one launch and one round trip of the shuffled tensor less).  VLLM_PS_FOLD=0: A/B.
    const char *ps_env = getenv("VLLM_PS_FOLD");   // (read per call: the test compares both forms in one process)
    const int ps_fold_off = ps_env && ps_env[0] == '0' ? 1 : 0;
    const bool ps_in_ln = d->pixel_shuffle && d->kind == VLLM_BRIDGE_INTERNVL_MLP && !ps_fold_off;
    NormGather psg = {0, 0, 0, 0};
    if (d->pixel_shuffle) {
        const int hw = (int)(sqrtf((float)T_in) + 0.5f);
        VLLM_REQUIRE(hw * hw == T_in && hw % 2 == 0, "bridge: pixel_shuffle needs an even square token grid (T=%d)", T_in);
        if (ps_in_ln) {
            psg.hw = hw; psg.tok0 = d->skip_cls ? 1 : 0; psg.cseg = C / 8; psg.tile_stride = (long)S * C;
        } else {
            uint16_t *shuf = (uint16_t *)(ws + w.a);
            prof_mark(PT_BRIDGE_OTHER, st);
            TRY(pixel_shuffle_launch(hidden, (long)S * C, C, d->skip_cls ? 1 : 0, shuf, n, hw, C, st));
            x = shuf; ldx = Cin;
        }
        xP = 0; rows = (long)n * (T_in / 4);
    }
    if (d->kind == VLLM_BRIDGE_INTERNVL_MLP) {
        VLLM_REQUIRE(d->ln_w && d->ln_b, "bridge: internvl_mlp needs LayerNorm parameters");
        uint16_t *ln = (uint16_t *)(ws + w.b);
        prof_mark(PT_BRIDGE_OTHER, st);
        if (ps_in_ln) {
            TRY(norm_bf16_launch(false, hidden, C, d->ln_w, d->ln_b, ln, Cin, rows, Cin, d->ln_eps, st, nullptr, 1, &psg));
        } else if (xP == 0) {
            TRY(norm_bf16_launch(false, x, ldx, d->ln_w, d->ln_b, ln, Cin, rows, Cin, d->ln_eps, st));
        } else {
            // no pixel-shuffle (modeling_visionllmv2.py:163-172 allows it): the LayerNorm reads hidden[:, 1:] in place, one
            // launch per tile (its rows are contiguous behind the tile's CLS row), and writes the rows densely
            for (int t = 0; t < n; ++t)
                TRY(norm_bf16_launch(false, hidden + ((long)t * S + 1) * C, C, d->ln_w, d->ln_b, ln + (long)t * T_in * Cin, Cin, T_in,
                                     Cin, d->ln_eps, st));
            xP = 0;
        }
        x = ln; ldx = Cin;
    }
    uint16_t *tmp[2] = {(uint16_t *)(ws + w.c), (uint16_t *)(ws + w.c + align256(rows * (long)Cout * 2))};
    void *sk = w.sk_bytes ? ws + w.sk : nullptr;
    if (sk) VLLM_REQUIRE(hipMemsetAsync(sk, 0, SK_FLAG_BYTES, st) == hipSuccess, "bridge: flag reset failed");
    int K = Cin;
    for (int i = 0; i < d->depth; ++i) {
        VLLM_REQUIRE(d->w[i], "bridge: weight %d missing", i);
        const bool last = i == d->depth - 1;
        uint16_t *y = last ? out : tmp[i & 1];
        // GELU sits between Linear i and Linear i+1 -> fused into Linear i's epilogue
        prof_mark(PT_BRIDGE_GEMM, st);
        TRY(gemm(st, last ? EPI_BIAS : EPI_GELU, x, ldx, d->w[i], K, d->b[i], y, Cout, (int)rows, Cout, K, nullptr, nullptr,
                 0, 0, i == 0 ? xP : 0, sk, w.sk_bytes));
        x = y; ldx = Cout; K = Cout;
    }
    prof_mark(PT_END, st);
    return VLLM_OK;
}
