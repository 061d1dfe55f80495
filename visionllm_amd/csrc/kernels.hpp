// Internal launch interface shared by the kernel files and the orchestrators (vit.cpp, bridge).
#pragma once
#include "common.hpp"

namespace vllm {

enum GemmEpilogue : int {
    EPI_BIAS = VLLM_EPI_BIAS,
    EPI_GELU = VLLM_EPI_GELU,
    EPI_QUICK_GELU = VLLM_EPI_QUICK_GELU,
    EPI_RESIDUAL = VLLM_EPI_RESIDUAL,
    EPI_EMBED = VLLM_EPI_EMBED,
    EPI_F32 = VLLM_EPI_F32,
    EPI_MSDA = 100,   // internal (msda_layer.hip): offsets -> sampling locations, logits -> softmax weights, both fp32
};

struct GemmArgs {
    const uint16_t *X;      // [M, K] row stride ldx
    const uint16_t *W;      // [N, K] row stride ldw
    uint16_t *Y;            // [M', N] row stride ldy
    const uint16_t *bias;   // [N] or null
    const uint16_t *scale;  // [N] layer-scale (EPI_RESIDUAL) or null
    const uint16_t *res;    // [M, N] residual, row stride ldr (EPI_RESIDUAL); [P+1, N] position table (EPI_EMBED)
    int M, N, K;
    int ldx, ldw, ldy, ldr;
    int P;                  // patches per image (EPI_EMBED)
    int mt, nt;             // tile counts (filled by the launcher)
    int xP;                 // >0: X is [n, 1+xP, K] and row m reads X row m + m/xP + 1 (CLS rows skipped)
    int variant;            // 0 auto, 1 force 128x128 kernel, 2 force 256x256 8-phase kernel (tuning / tests)
    int variant256;         // 8-phase kernel block rows: 0 auto, 3 -> 192, 4 -> 256
    int half_tail = 0;      // persistent schedule (launcher): 1 = the tiles of the last, incomplete round run as two half-height tiles each
    int direct_store;       // 8-phase epilogue: 0 through LDS, 1 straight from the accumulator layout, 2 automatic
    // EPI_MSDA only: output features [0, nsplit) come from W / bias and become sampling locations in Y (fp32, ldy),
    // features [nsplit, N) come from W2 / bias2 and become the per-head softmax over L*P == 16 logits in Y2 (fp32, ldy2)
    const uint16_t *W2 = nullptr, *bias2 = nullptr;
    float *Y2 = nullptr;
    const float *ref = nullptr;       // [M, mL, ref_dim] reference points
    const int64_t *shapes = nullptr;  // [mL, 2] (H, W)
    int nsplit = 0, ldy2 = 0, mL = 0, mP = 0, ref_dim = 0, four_d = 0;
    const uint8_t *row_mask = nullptr;   // EPI_BIAS on the streaming kernel (gemm_skinny.hip; msda_layer.hip's bf16 value): rows with a non-zero byte are stored as zeros
    int res_init = 0;       // EPI_RESIDUAL without LayerScale, 8-phase kernel: the residual is the accumulators' initial value (launcher)
    int prof = 0;           // VLLM_GEMM_PROF=1: the 8-phase kernel adds prologue / main loop / epilogue ticks to device counters
    // Stream-K tail of the 8-phase kernel (gemm256.hip): scratch for fp32 partial tiles + one flag per scratch slot, provided by
    // the caller (the orchestrators carve them out of their workspace; flags zeroed once per forward).  The launcher fills sk_*.
    float *sk_ws = nullptr;
    unsigned *sk_flags = nullptr;
    long sk_ws_bytes = 0;
    int sk_tiles = 0;       // tiles (the last ones of the dense order) whose K iterations are spread evenly over sk_blocks blocks
    int sk_dp = 0;          // tiles in front of them: one block each, as without stream-K
    int sk_blocks = 0;
    // LayerNorm / RMSNorm folded into the GEMMs around it (8-phase kernel, row-wise LDS epilogue; vit.cpp):
    //   the PRODUCER of the normalised tensor (proj / fc2 + residual) leaves, per output row and 256-column tile, {mean, M2} (or
    //   {sum of squares, -} for RMSNorm) of the bf16 values it stores: ln_out [M][nt][2];
    //   the CONSUMER (qkv / fc1) reads the UN-normalised rows as its X operand, multiplies with weights that carry the norm's
    //   gamma, and applies the row statistics in its epilogue:  y = act(r_m * acc + (-r_m * mean_m) * colsum_n + bias'_n).
    float *ln_out = nullptr;
    const float *ln_in = nullptr;
    int ln_slots = 0;               // column tiles per row of ln_in
    int ln_cols = 0;                // elements per normalised row (the producer's N)
    int ln_rms = 0;                 // 1: RMSNorm (second moment about zero, no shift)
    int dry_run = 0;                // 1: gemm256_bf16_launch only answers whether the persistent schedule would take the call (VLLM_OK) or not
    int ln_wide = 0;                // 1 (round 5, RMSNorm only, the persistent kernel only): statistics as [M][16] floats -- slot s = sum of squares of
                                    //    column tile s, unused slots 0 -- for rows of up to 16 column tiles (hidden 3200: 13); 0: [M][nt][2], four tiles
    float ln_eps = 0.f;
    const float *ln_colsum = nullptr;   // consumer: s_n = sum_k W'[n, k] (LayerNorm), fp32 [N]
    const float *ln_bias = nullptr;     // consumer: bias'_n = b_n + sum_k beta_k W[n, k] (fp32 [N]; NULL = 0); replaces `bias`
    // consumer, filled by the launcher from ln_cols (uniform float arithmetic in the kernel would sit in VECTOR registers across
    // the main loop: that is what spilled): Chan's update of the running {mean, M2} with column tile s is
    //   mean += (mean_s - mean) * ln_cw[s];  M2 += M2_s + (mean_s - mean)^2 * ln_cc[s];   ln_inv_cols = 1 / ln_cols
    float ln_cw[4] = {0.f, 0.f, 0.f, 0.f}, ln_cc[4] = {0.f, 0.f, 0.f, 0.f}, ln_inv_cols = 0.f;
    int no_persist = 0;             // 1: one workgroup per tile even where the persistent schedule (gemm256p.hip) would take the GEMM
    int tile_rb = 0;                // persistent schedule, tile order: 0 = the dense XCD order of gemm256.hip; RB > 0 = banded (launcher)
    unsigned long long *trace = nullptr;   // VLLM_GEMM_TRACE=<device address of 3 x 8192 uint64>: per block {start, end} in
                                           // 100 MHz s_memrealtime ticks + HW_ID (which CU), for tools/prof_gemm256.py
};

int gemm_direct_store();       // VLLM_GEMM_DIRECT_STORE / vllm_set_option("gemm_direct_store")
int gemm_variant_override();   // VLLM_GEMM_VARIANT / vllm_set_option("gemm_variant")
extern int g_gemm_tile_rb;
int gemm_tile_rb();            // VLLM_GEMM_TILE_RB / vllm_set_option("gemm_tile_rb"): -1 automatic, 0 dense XCD order, RB > 0 banded
int attn_variant();            // VLLM_ATTN_VARIANT / vllm_set_option("attn_variant"): bit0 pipe, bit1 defer, bit2 prio,
                               // bit3 asm tr-reads, bit4 no padding trim, 32 = automatic (default)
// gemm_skinny.hip: K = 256, N = 256 / 384 (the linears of a deformable-attention layer); VLLM_GEMM_SKINNY / option "gemm_skinny"
bool gemm_skinny_takes(int epi, const GemmArgs &a);
int gemm_skinny_launch(int epi, const GemmArgs &a, hipStream_t st);
int gemm_skinny_enabled();
int gemm_skinny_set(int v);
struct Dcnv3Geo;
bool dcnv3_bwd_mfma_takes(const Dcnv3Geo &q);   // msda_bwd_mfma.hip (template flag DCN): fp32, group channels 32
int dcnv3_bwd_mfma_launch(const float *input, const float *offset, const float *mask, const float *grad_out, const Dcnv3Geo &q,
                          float offset_scale, float *grad_input, float *grad_offset, float *grad_mask, hipStream_t st);
int dcnv3_bwd_tiled();       // VLLM_DCNV3_BWD_TILED / vllm_set_option("dcnv3_bwd_tiled")
int dcnv3_bwd_tiled_set(int v);
int gemm_half_tail();        // VLLM_GEMM_HALF_TAIL / vllm_set_option("gemm_half_tail"): half-height tiles in the persistent GEMM's last round
int gemm_half_tail_set(int v);
long gemm256p_half_launches();
int msda_layer_value_bf16_set(int v);   // VLLM_MSDA_LAYER_VALUE_BF16 / vllm_set_option("msda_layer_value_bf16")
int msda_layer_fused();        // VLLM_MSDA_LAYER_FUSED / vllm_set_option("msda_layer_fused")
int msda_tiled_enabled();      // VLLM_MSDA_TILED / vllm_set_option("msda_tiled")

int gemm_bf16_launch(int epi, GemmArgs a, hipStream_t st);

// In-step kernel timing (vllm_prof_enable / vllm_prof_read): when enabled, the orchestrators record a HIP event in front of
// every operator they enqueue; the time from one mark to the next is attributed to the first one's tag, so a kernel is
// charged its duration INSIDE the step (queueing behind its predecessor and the gap to its successor included) -- what a
// step costs, not what the kernel does alone.  Disabled (the default) a mark is one load and a branch.
enum ProfTag : int { PT_END = 0, PT_EMBED, PT_NORM, PT_QKV, PT_QKNORM, PT_ATTN, PT_PROJ, PT_FC1, PT_FC2, PT_BRIDGE_GEMM, PT_BRIDGE_OTHER,
                     PT_MSDA_ENC, PT_MSDA_OTHER, PT_MSDA_LAYER, PT_COUNT };
extern int g_prof_on;
void prof_mark_slow(int tag, hipStream_t st);
inline void prof_mark(int tag, hipStream_t st) { if (g_prof_on) prof_mark_slow(tag, st); }

// Scratch of the 8-phase GEMM's stream-K tail: [4 KiB of flags, zero before the first use][SK_MAX_BLOCKS slots of 256 KiB].
// The orchestrators reserve it in their workspace (and zero the flags once per forward); vllm_gemm_bf16_sk takes it from the caller.
// One slot per stream-K block = per compute unit of the device the launch runs on; a device with more than SK_MAX_BLOCKS CUs (or a
// caller's smaller buffer) simply does not get the stream-K route: gemm256_launch checks sk_ws_bytes >= cus * slot before planning it.
constexpr long SK_FLAG_BYTES = 4096, SK_SLOT_BYTES = 32L * 512 * 16, SK_MAX_BLOCKS = 320;
constexpr long SK_SCRATCH_BYTES = SK_FLAG_BYTES + SK_MAX_BLOCKS * SK_SLOT_BYTES;
inline void gemm_set_scratch(GemmArgs &a, void *scratch, long bytes)
{
    if (!scratch || bytes < SK_FLAG_BYTES + SK_SLOT_BYTES) return;
    a.sk_flags = reinterpret_cast<unsigned *>(scratch);
    a.sk_ws = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + SK_FLAG_BYTES);
    a.sk_ws_bytes = bytes - SK_FLAG_BYTES;
}

inline int gemm(hipStream_t st, int epi, const uint16_t *X, int ldx, const uint16_t *W, int ldw, const uint16_t *bias,
                uint16_t *Y, int ldy, int M, int N, int K, const uint16_t *scale = nullptr, const uint16_t *res = nullptr,
                int ldr = 0, int P = 0, int xP = 0, void *scratch = nullptr, long scratch_bytes = 0)
{
    GemmArgs a;
    gemm_set_scratch(a, scratch, scratch_bytes);
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.scale = scale; a.res = res;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.P = P; a.mt = a.nt = 0; a.xP = xP; a.variant = gemm_variant_override(); a.variant256 = 0; a.direct_store = gemm_direct_store();
    return gemm_bf16_launch(epi, a, st);
}

// G > 1: G column groups of C elements per memory row, normalised independently in ONE launch (G = 2: the q and k blocks of
// a qkv row with weights w / w2 -- InternViT's q_norm + k_norm, modeling_intern_vit.py:131-134)
// Pixel-shuffled input rows for the norm kernel (round 5: the InternVL projector's LayerNorm reads the 2 x 2 token neighbourhoods of
// hidden[:, 1:] itself -- modeling_visionllmv2.py:381-392, 574-579 -- instead of a pixel-shuffle launch writing them first): output row
// ((n * h2 + i2) * h2 + j2) is the concatenation over quad = 2 a + b of token tok0 + (2 i2 + a) * hw + (2 j2 + b) of tile n, cseg
// 16-byte chunks each.
struct NormGather {
    int hw, tok0, cseg;
    long tile_stride;   // elements between two tiles of the input
};
int norm_bf16_launch(bool rms, const uint16_t *x, int ldx, const uint16_t *w, const uint16_t *b, uint16_t *y, int ldy,
                     long rows, int C, float eps, hipStream_t st, const uint16_t *w2 = nullptr, int G = 1, const NormGather *ps = nullptr);

struct AttnArgs {
    const uint16_t *q, *k, *v;  // element pointers; D contiguous
    uint16_t *out;              // [B, S, H, D]
    long q_bs, k_bs, v_bs;      // batch strides (elements)
    int q_ts, k_ts, v_ts;       // token strides
    int q_hs, k_hs, v_hs;       // head strides
    int B, S, H;
    int nqt;                    // query tiles per (b, h) (filled by the launcher)
    int nq = 1;                 // schedule 2: consecutive query tiles one block processes (filled by the launcher)
    int stagger = 0, n_cu = 256;   // schedule 2: start skew between the blocks of a CU [cycles], CUs of the device (launcher)
    int row0;                   // first query row of this launch (filled by the launcher)
    int kx = 0, qx = 0;         // round 6: leading tokens taken out of the key tiling (initial softmax state) / the query tiling
    int cls_wave = 0;           // > 0: the 32-row group index that holds query row 0 alone (the spare wave of the last query block)
    int no_trim;                // 1: process padding keys / padding query waves like live ones (A/B switch; launcher)
    float scale_log2e;
    int f16 = 0;                // 1: q / k / v / out are IEEE half (flash_attention.py:39-41 accepts fp16 and bf16); fp32 softmax either way
};
int attn_fwd_launch(AttnArgs a, int D, hipStream_t st);

int im2col_launch(const void *pixels, int pixel_is_f32, uint16_t *A, int N, int img, int ps, int Kpad, hipStream_t st);
int cls_rows_launch(const uint16_t *cls, const uint16_t *pos, uint16_t *hidden, int N, int S, int C, hipStream_t st);
int pixel_shuffle_launch(const uint16_t *in, long in_tile_stride, int ldin, int tok0, uint16_t *out, int N, int hw,
                         int C, hipStream_t st);

}  // namespace vllm
