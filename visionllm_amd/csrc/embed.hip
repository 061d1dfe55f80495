// Data-movement kernels around the ViT GEMMs (HBM-bound gathers, coalesced on the write side).
//
//   im2col_patches : pixels [N,3,IMG,IMG] -> A [N*P, Kpad] with k = c*ps*ps + ky*ps + kx (== Conv2d weight
//                    flattening), zero padded to Kpad (multiple of 64) so the patch embedding runs as a GEMM.
//                    Replaces the strided gather inside nn.Conv2d(3, C, k=s=14)
//                    (VisionLLMv2/visionllmv2/model/internvit/modeling_intern_vit.py:73-75, 85-86).
//   cls_rows       : hidden[n, 0, :] = class_embedding + position_embedding[0]   (:87-89)
//   pixel_shuffle  : [N, 1+h*w, C] (CLS dropped) -> [N, (h/2)*(w/2), 4C] with the InternVL (w,h) permute order
//                    (visionllmv2/model/modeling_visionllmv2.py:381-392, 569-578)
//   drop_cls       : [N, 1+T, C] -> [N, T, C]   (:571)  (only used by the stand-alone bridge entry point; the
//                    fused path lets the projector GEMM skip the CLS rows while staging)
#include "common.hpp"
#include "kernels.hpp"

namespace vllm {

// One thread per (patch, channel, ky) row segment of `ps` pixels.
template <typename PIX>
__global__ __launch_bounds__(256) void im2col_kernel(const PIX *__restrict__ px, uint16_t *__restrict__ A, int N,
                                                     int img, int ps, int g /*patches per side*/, int Kpad)
{
    const int P = g * g;
    const long nseg = (long)N * P * 3 * ps;
    const int K = 3 * ps * ps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nseg; i += (long)gridDim.x * blockDim.x) {
        const int ky = (int)(i % ps);
        long t = i / ps;
        const int c = (int)(t % 3);
        t /= 3;
        const int p = (int)(t % P);
        const int n = (int)(t / P);
        const int pyy = p / g, pxx = p % g;
        const PIX *src = px + (((long)n * 3 + c) * img + (pyy * ps + ky)) * img + pxx * ps;
        uint16_t *dst = A + ((long)n * P + p) * Kpad + c * ps * ps + ky * ps;
        for (int kx = 0; kx < ps; ++kx) {
            if constexpr (sizeof(PIX) == 2) dst[kx] = (uint16_t)src[kx];
            else dst[kx] = f32_to_bf16((float)src[kx]);
        }
        if (c == 2 && ky == ps - 1)
            for (int k = K; k < Kpad; ++k) A[((long)n * P + p) * Kpad + k] = 0;
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(const uint16_t *__restrict__ cls, const uint16_t *__restrict__ pos,
                                                       uint16_t *__restrict__ hidden, int N, int S, int C)
{
    const long n_el = (long)N * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long n = i / C;
        // reference adds in the parameter dtype (bf16): one rounding of the sum
        hidden[n * (long)S * C + c] = f32_to_bf16(bf16_to_f32(cls[c]) + bf16_to_f32(pos[c]));
    }
}

// 16-byte chunk copy: out[n][i2*(h/2)+j2][a*2C + bs*C + ch] = in[n][1 + (2*i2+a)*h + (2*j2+bs)][ch]
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const uint16_t *__restrict__ in, long in_tile_stride,
                                                            int ldin, int tok0, uint16_t *__restrict__ out, int N, int hw,
                                                            int C)
{
    const int h2 = hw / 2, cch = C / 8;
    const long nchunks = (long)N * h2 * h2 * 4 * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        const int ck = (int)(i % cch);
        long t = i / cch;
        const int quad = (int)(t % 4);
        t /= 4;
        const int j2 = (int)(t % h2);
        t /= h2;
        const int i2 = (int)(t % h2);
        const long n = t / h2;
        const int a = quad >> 1, bs = quad & 1;
        const long tok = tok0 + (long)(2 * i2 + a) * hw + (2 * j2 + bs);
        const uint4_t v = *reinterpret_cast<const uint4_t *>(in + n * in_tile_stride + tok * ldin + ck * 8);
        *reinterpret_cast<uint4_t *>(out + (((n * h2 + i2) * h2 + j2) * 4L + quad) * C + ck * 8) = v;
    }
}

__global__ __launch_bounds__(256) void drop_cls_kernel(const uint16_t *__restrict__ in, long in_tile_stride, int ldin,
                                                       uint16_t *__restrict__ out, int N, int T, int C)
{
    const int cch = C / 8;
    const long nchunks = (long)N * T * cch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        const int ck = (int)(i % cch);
        long t = i / cch;
        const int tok = (int)(t % T);
        const long n = t / T;
        *reinterpret_cast<uint4_t *>(out + (n * T + tok) * (long)C + ck * 8) =
            *reinterpret_cast<const uint4_t *>(in + n * in_tile_stride + (long)(1 + tok) * ldin + ck * 8);
    }
}

// Row movers.  Round 5: a WAVE owns a row (its index is one scalar-cached load, no per-chunk 64-bit division -- the round 1-4 form
// spent ~80 instructions of integer division per 16-byte chunk and ran at 0.18 of the HBM roofline) and its lanes walk the row's
// 16-byte chunks, four loads in flight per lane.
__device__ __forceinline__ void move_row(const uint16_t *__restrict__ s, uint16_t *__restrict__ d, int cch, int lane)
{
    int ck = lane;
    for (; ck + 192 < cch; ck += 256) {
        const uint4_t a = *reinterpret_cast<const uint4_t *>(s + ck * 8), b = *reinterpret_cast<const uint4_t *>(s + (ck + 64) * 8);
        const uint4_t c = *reinterpret_cast<const uint4_t *>(s + (ck + 128) * 8), e = *reinterpret_cast<const uint4_t *>(s + (ck + 192) * 8);
        *reinterpret_cast<uint4_t *>(d + ck * 8) = a; *reinterpret_cast<uint4_t *>(d + (ck + 64) * 8) = b;
        *reinterpret_cast<uint4_t *>(d + (ck + 128) * 8) = c; *reinterpret_cast<uint4_t *>(d + (ck + 192) * 8) = e;
    }
    for (; ck < cch; ck += 64) *reinterpret_cast<uint4_t *>(d + ck * 8) = *reinterpret_cast<const uint4_t *>(s + ck * 8);
}

// dst[idx[i], :] = src[i, :]   (16-byte chunks; idx int64 on the device)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint16_t *__restrict__ src, const int64_t *__restrict__ idx,
                                                           uint16_t *__restrict__ dst, long n, int C, long dst_rows)
{
    const int cch = C / 8, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long r = wave; r < n; r += nwaves) {
        const long d = idx[r];
        if (d >= 0 && d < dst_rows) move_row(src + r * C, dst + d * C, cch, lane);
    }
}

// dst[di[i], :] = src[si[i], :]; a null index means the identity.  Rows of C bf16.
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint16_t *__restrict__ src, const int64_t *__restrict__ si,
                                                        uint16_t *__restrict__ dst, const int64_t *__restrict__ di, long n, int C,
                                                        long src_rows, long dst_rows)
{
    const int cch = C / 8, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long r = wave; r < n; r += nwaves) {
        const long s = si ? si[r] : r, d = di ? di[r] : r;
        if (s >= 0 && s < src_rows && d >= 0 && d < dst_rows) move_row(src + s * C, dst + d * C, cch, lane);
    }
}

// ---- visual-token splice with the index bookkeeping on the device (round 5; modeling_visionllmv2.py:582-605) ----------------------
// The reference selects with boolean masks (`image_features[has_image]`, `inputs_embeds[selected] = ...`): two host synchronisations
// and two full copies.  Here one block lists the <im_patch> slots in order (chunked count + block scan), marks the samples that have
// an image, lists THEIR tiles, and decides the token-count rule of :597-603 (equal, or slots a whole multiple of the tokens: repeated;
// anything else: nothing is written and the error flag is raised); a second kernel moves a row per wave.
// Workspace (int32): [0] rows to move, [1] visual tokens offered, [2] error, [3] slots found, [4 ...) slot positions [B L],
// then the kept tiles [n_tiles], then (B > 512 only) the tiles per sample [B].
constexpr int SPL_THREADS = 1024, SPL_HDR = 4, SPL_MAXB = 4096, SPL_BYVAL = 512;
struct SplTiles { int n[SPL_BYVAL]; };   // tiles per sample by value (a kernel argument: no host-to-device copy) for B <= 512

__global__ __launch_bounds__(SPL_THREADS) void splice_index_kernel(const int64_t *__restrict__ ids, long imp_id, const SplTiles tv,
                                                                   const int32_t *__restrict__ tiles_dev, int tiles_mode, int B, int L,
                                                                   int n_tiles, int T, int32_t *__restrict__ ws,
                                                                   int32_t *__restrict__ status)
{
    __shared__ int s_cnt[SPL_THREADS / 64], s_has[SPL_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = SPL_THREADS / 64;
    const long n = (long)B * L;
    // a wave owns a contiguous segment, read 64 positions (512 contiguous bytes) at a time
    const long per = ((n + NW - 1) / NW + 63) & ~63L, lo = wave * per, hi = lo + per < n ? lo + per : n;
    for (int b = tid; b < B; b += SPL_THREADS) s_has[b] = 0;
    __syncthreads();
    int c = 0;
    for (long i0 = lo; i0 < hi; i0 += 64) {
        const long i = i0 + lane;
        const bool sel = i < hi && ids[i] == imp_id;
        if (sel) s_has[i / L] = 1;   // (benign race: every writer stores 1)
        c += __popcll(__ballot(sel));
    }
    if (lane == 0) s_cnt[wave] = c;
    __syncthreads();
    int r = 0, n_sel = 0;
    for (int w = 0; w < NW; ++w) { const int v = s_cnt[w]; r += w < wave ? v : 0; n_sel += v; }
    int32_t *slots = ws + SPL_HDR, *kept = ws + SPL_HDR + n;
    for (long i0 = lo; i0 < hi; i0 += 64) {   // second pass over the (cached) ids: ranks from the ballot
        const long i = i0 + lane;
        const bool sel = i < hi && ids[i] == imp_id;
        const unsigned long long m = __ballot(sel);
        if (sel) slots[r + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
        r += __popcll(m);
    }
    if (tid == 0) {
        int k = 0, t0 = 0, bad = 0;
        for (int b = 0; b < B; ++b) {   // the tiles of the samples that have an image, in order (:585-592)
            const int nt = tiles_mode == 0 ? 1 : tiles_mode == 1 ? tv.n[b] : tiles_dev[b];
            if (nt < 0 || t0 + nt > n_tiles) { bad = 1; break; }
            if (s_has[b])
                for (int j = 0; j < nt; ++j) kept[k++] = t0 + j;
            t0 += nt;
        }
        if (t0 != n_tiles) bad = 1;
        const long n_vit = (long)k * T;
        if (!bad && n_sel != n_vit && !(n_vit > 0 && n_sel > n_vit && n_sel % n_vit == 0)) bad = 1;
        ws[0] = bad ? 0 : n_sel; ws[1] = (int32_t)n_vit; ws[2] = bad; ws[3] = n_sel;
        if (status) { status[0] = n_sel; status[1] = (int32_t)n_vit; status[2] = bad; status[3] = k; }
    }
}

// row i of the slot list <- visual token i (mod the tokens offered) of the kept tiles; a wave per row
__global__ __launch_bounds__(256) void splice_move_kernel(const uint16_t *__restrict__ feats, const int32_t *__restrict__ ws, long n,
                                                          int T, int C, uint16_t *__restrict__ embeds)
{
    const int cch = C / 8, lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int n_move = ws[0], n_vit = ws[1];
    const int32_t *slots = ws + SPL_HDR, *kept = ws + SPL_HDR + n;
    for (long r = wave; r < n_move; r += nwaves) {
        const int j = (int)(r % n_vit), kt = j / T, tok = j - kt * T;
        move_row(feats + ((long)kept[kt] * T + tok) * C, embeds + (long)slots[r] * C, cch, lane);
    }
}

static inline unsigned grid_for(long n)
{
    long b = (n + 255) / 256;
    if (b > 256L * 16) b = 256L * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int im2col_launch(const void *pixels, int pixel_is_f32, uint16_t *A, int N, int img, int ps, int Kpad, hipStream_t st)
{
    VLLM_REQUIRE(pixels && A, "im2col: null pointer");
    VLLM_REQUIRE(ps > 0 && img % ps == 0 && Kpad >= 3 * ps * ps, "im2col: bad geometry img=%d patch=%d Kpad=%d", img, ps, Kpad);
    if (N == 0) return VLLM_OK;
    const int g = img / ps;
    const long nseg = (long)N * g * g * 3 * ps;
    if (pixel_is_f32)
        VLLM_LAUNCH((im2col_kernel<float>), dim3(grid_for(nseg)), dim3(256), 0, st, (const float *)pixels, A, N, img, ps, g, Kpad);
    else
        VLLM_LAUNCH((im2col_kernel<uint16_t>), dim3(grid_for(nseg)), dim3(256), 0, st, (const uint16_t *)pixels, A, N, img, ps, g, Kpad);
    VLLM_CHECK_LAUNCH("im2col_kernel");
    return VLLM_OK;
}

int cls_rows_launch(const uint16_t *cls, const uint16_t *pos, uint16_t *hidden, int N, int S, int C, hipStream_t st)
{
    if (N == 0) return VLLM_OK;
    VLLM_LAUNCH(cls_rows_kernel, dim3(grid_for((long)N * C)), dim3(256), 0, st, cls, pos, hidden, N, S, C);
    VLLM_CHECK_LAUNCH("cls_rows_kernel");
    return VLLM_OK;
}

int pixel_shuffle_launch(const uint16_t *in, long in_tile_stride, int ldin, int tok0, uint16_t *out, int N, int hw,
                         int C, hipStream_t st)
{
    VLLM_REQUIRE(in && out, "pixel_shuffle: null pointer");
    VLLM_REQUIRE(hw % 2 == 0 && C % 8 == 0 && ldin % 8 == 0 && in_tile_stride % 8 == 0 && aligned16(in) && aligned16(out),
                 "pixel_shuffle: needs even grid, C %% 8 == 0 and 16-byte alignment");
    if (N == 0) return VLLM_OK;
    VLLM_LAUNCH(pixel_shuffle_kernel, dim3(grid_for((long)N * hw * hw * C / 8)), dim3(256), 0, st, in,
                       in_tile_stride, ldin, tok0, out, N, hw, C);
    VLLM_CHECK_LAUNCH("pixel_shuffle_kernel");
    return VLLM_OK;
}

int drop_cls_launch(const uint16_t *in, long in_tile_stride, int ldin, uint16_t *out, int N, int T, int C, hipStream_t st)
{
    VLLM_REQUIRE(in && out && C % 8 == 0 && ldin % 8 == 0 && in_tile_stride % 8 == 0 && aligned16(in) && aligned16(out),
                 "drop_cls: bad arguments");
    if (N == 0) return VLLM_OK;
    VLLM_LAUNCH(drop_cls_kernel, dim3(grid_for((long)N * T * C / 8)), dim3(256), 0, st, in, in_tile_stride, ldin,
                       out, N, T, C);
    VLLM_CHECK_LAUNCH("drop_cls_kernel");
    return VLLM_OK;
}

}  // namespace vllm

using namespace vllm;

extern "C" int vllm_scatter_rows_bf16(const uint16_t *src, const int64_t *idx, uint16_t *dst, long n, int C, long dst_rows,
                                      vllm_stream_t stream)
{
    VLLM_REQUIRE(n >= 0 && C > 0 && C % 8 == 0, "scatter_rows: C must be a positive multiple of 8");
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(src && idx && dst && aligned16(src) && aligned16(dst), "scatter_rows: null or unaligned pointer");
    VLLM_LAUNCH(scatter_rows_kernel, dim3(grid_for(n * 64)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, C,
                dst_rows);
    VLLM_CHECK_LAUNCH("scatter_rows_kernel");
    return VLLM_OK;
}

extern "C" long vllm_splice_workspace_ints(int B, int L, int n_tiles) { return (long)SPL_HDR + (long)B * L + n_tiles + (B > SPL_BYVAL ? B : 0); }

extern "C" int vllm_splice_visual_tokens_bf16(const int64_t *input_ids, long imp_token_id, const uint16_t *image_features,
                                              const int32_t *tiles_per_sample, int B, int L, int n_tiles, int T, int C,
                                              uint16_t *inputs_embeds, int32_t *workspace, int32_t *status, vllm_stream_t stream)
{
    VLLM_REQUIRE(B >= 0 && L >= 0 && n_tiles >= 0 && T >= 0 && C > 0 && C % 8 == 0, "splice_visual_tokens: bad sizes (C must be a positive multiple of 8)");
    VLLM_REQUIRE(B <= SPL_MAXB && (long)B * L < (1L << 31) && (long)n_tiles * T < (1L << 31), "splice_visual_tokens: too many samples / positions / tokens for one call");
    VLLM_REQUIRE(workspace, "splice_visual_tokens: null workspace");
    VLLM_REQUIRE((long)B * L == 0 || (input_ids && inputs_embeds && aligned16(inputs_embeds)), "splice_visual_tokens: null or unaligned pointer");
    VLLM_REQUIRE((long)n_tiles * T == 0 || (image_features && aligned16(image_features)), "splice_visual_tokens: null or unaligned image_features");
    SplTiles tv;
    int mode = 0;
    const int32_t *tdev = nullptr;
    if (tiles_per_sample && B <= SPL_BYVAL) {
        mode = 1;
        for (int b = 0; b < B; ++b) tv.n[b] = tiles_per_sample[b];
    } else if (tiles_per_sample) {   // (more than 512 samples: through the workspace)
        mode = 2;
        int32_t *d = workspace + SPL_HDR + (long)B * L + n_tiles;
        VLLM_REQUIRE(hipMemcpyAsync(d, tiles_per_sample, sizeof(int32_t) * B, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess,
                     "splice_visual_tokens: copy of the tiles per sample failed");
        tdev = d;
    }
    VLLM_LAUNCH(splice_index_kernel, dim3(1), dim3(SPL_THREADS), 0, (hipStream_t)stream, input_ids, imp_token_id, tv, tdev, mode, B, L,
                n_tiles, T, workspace, status);
    VLLM_CHECK_LAUNCH("splice_index_kernel");
    if ((long)B * L == 0 || (long)n_tiles * T == 0) return VLLM_OK;
    VLLM_LAUNCH(splice_move_kernel, dim3(grid_for((long)B * L * 64)), dim3(256), 0, (hipStream_t)stream, image_features, workspace,
                (long)B * L, T, C, inputs_embeds);
    VLLM_CHECK_LAUNCH("splice_move_kernel");
    return VLLM_OK;
}

extern "C" int vllm_copy_rows_bf16(const uint16_t *src, const int64_t *src_idx, uint16_t *dst, const int64_t *dst_idx, long n,
                                   int C, long src_rows, long dst_rows, vllm_stream_t stream)
{
    VLLM_REQUIRE(n >= 0 && C > 0 && C % 8 == 0, "copy_rows: C must be a positive multiple of 8");
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(src && dst && aligned16(src) && aligned16(dst), "copy_rows: null or unaligned pointer");
    VLLM_LAUNCH(copy_rows_kernel, dim3(grid_for(n * 64)), dim3(256), 0, (hipStream_t)stream, src, src_idx, dst, dst_idx, n, C,
                src_rows, dst_rows);
    VLLM_CHECK_LAUNCH("copy_rows_kernel");
    return VLLM_OK;
}

extern "C" int vllm_im2col_patches(const void *pixels, int pixel_is_f32, uint16_t *A, int N, int img, int patch,
                                   int Kpad, vllm_stream_t stream)
{
    return im2col_launch(pixels, pixel_is_f32, A, N, img, patch, Kpad, (hipStream_t)stream);
}

extern "C" int vllm_pixel_shuffle_bf16(const uint16_t *hidden, long tile_stride, int ld, int tok0, uint16_t *out, int N,
                                       int hw, int C, vllm_stream_t stream)
{
    return pixel_shuffle_launch(hidden, tile_stride, ld, tok0, out, N, hw, C, (hipStream_t)stream);
}
