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
