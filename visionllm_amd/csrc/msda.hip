// Multi-scale deformable attention (MSDA) for gfx950 -- forward (vectorised gather), sample-index probe, backward.
//
// Replaces the reference's ms_deformable_im2col_gpu_kernel / col2im kernels
// (VisionLLMv2/visionllmv2/model/unipose/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-298, 301-920; mmcv twin
// mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:201-255).  Semantics kept exactly:
//   h_im = loc_y*H - 0.5 (two roundings, never an FMA), accept iff h_im>-1 && w_im>-1 && h_im<H && w_im<W,
//   floor -> 4 corners, each corner contributes only when inside the map, bilinear weights hh*hw, hh*lw, lh*hw,
//   lh*lw, sample * attention weight accumulated over L*P points.
//
// MI355X design (not a translation -- the reference runs one thread per output scalar and re-reads loc/weight
// D times):
//   * one LPG-lane group per (b,q,m) "pair", LPG = D/CPL lanes, each lane owning CPL = 16 B worth of channels
//     (4 fp32 / 8 bf16): every corner gather is one coalesced 16 B/lane load; a wave64 holds 64/LPG CONSECUTIVE
//     QUERIES OF ONE HEAD (not one query x all heads): consecutive queries sample neighbouring pixels, so the
//     wave keeps re-touching the same 128-byte lines and the vector L1 absorbs most of the 18x gather
//     amplification (measured: the one-query-x-8-heads mapping was bound by L2 line requests, 1040 us @ cfg 4);
//   * sampling locations / weights of the wave's pairs (128 B per pair) are loaded ONCE per wave into a
//     wave-private LDS slice and re-read as LDS broadcasts;
//   * all 4*P corner loads of a level are issued unconditionally from clamped addresses (selects, not branches,
//     decide what contributes) so the memory pipeline sees 16 independent 16 B gathers per lane per level;
//   * level metadata (H, W, start) comes from the reference's DEVICE int64 tensors through scalar loads;
//   * blockIdx -> work mapping is XCD-aware: each XCD (blockIdx % 8) walks one contiguous 1/8 of the pairs, so
//     a feature map band lives in ONE XCD's 4 MiB L2 instead of eight.
// Roofline: HBM-bound on paper (11 flop/B); in practice limited by the 64 B/clk/CU vector-L1 path because the
// gathered bytes are 18x the compulsory bytes (DESIGN.md section "MSDA").
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

bool msda_tiled_ok(int D, int L, int P, int Lq, int S, int B, int M, const void *value, const void *out, const void *loc);
int msda_tiled_enabled();   // runtime.cpp
bool msda_tiled6_ok(int D, int L, int P, int Lq, int S, int B, int M);   // msda_tiled6.hip
int msda_tiled6_launch_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                            const float *attw, int B, int S, int M, int L, int Lq, uint16_t *out, hipStream_t st);
int msda_tiled_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *attw, int B, int S, int M, int L, int Lq, int P, float *out, hipStream_t st,
                      uint16_t *out16 = nullptr, int *wrote16 = nullptr, int geometry = 0);

// ---------------------------------------------------------------------------------------------------------
// Vectorised forward.
//   CPL = channels per lane (16 B), LPG = lanes per pair, PT = points per level (compile time: 1, 2, 4, 8).
// ---------------------------------------------------------------------------------------------------------
template <bool BF16>
struct ValueIO;

template <>
struct ValueIO<false> {  // fp32: 4 channels per 16 B
    static constexpr int CPL = 4;
    typedef float elem_t;
    __device__ static __forceinline__ void load(const float *p, float (&v)[4])
    {
        const float4_t t = *reinterpret_cast<const float4_t *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float *p, const float (&v)[4])
    {
        float4_t t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<float4_t *>(p) = t;
    }
};

template <>
struct ValueIO<true> {  // bf16: 8 channels per 16 B
    static constexpr int CPL = 8;
    typedef uint16_t elem_t;
    __device__ static __forceinline__ void load(const uint16_t *p, float (&v)[8])
    {
        const uint4_t t = *reinterpret_cast<const uint4_t *>(p);
        v[0] = bf16lo_to_f32(t.x); v[1] = bf16hi_to_f32(t.x);
        v[2] = bf16lo_to_f32(t.y); v[3] = bf16hi_to_f32(t.y);
        v[4] = bf16lo_to_f32(t.z); v[5] = bf16hi_to_f32(t.z);
        v[6] = bf16lo_to_f32(t.w); v[7] = bf16hi_to_f32(t.w);
    }
    __device__ static __forceinline__ void store(uint16_t *p, const float (&v)[8])
    {
        uint4_t t;
        t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]);
        t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4_t *>(p) = t;
    }
};

constexpr int MSDA_BLOCK = 256;          // 4 waves
constexpr int MSDA_WAVES = MSDA_BLOCK / 64;

template <bool BF16, int LPG, int PT>
__global__ __launch_bounds__(MSDA_BLOCK) void msda_fwd_vec_kernel(
    const typename ValueIO<BF16>::elem_t *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc, const float *__restrict__ attw,
    int S, int M, int L, int P, long n_bq /* B*Lq */, long Lq, long n_chunks,
    typename ValueIO<BF16>::elem_t *__restrict__ out, int skip_pyramid)
{
    if (skip_pyramid && geometry_is_pyramid(shapes, L, Lq)) return;   // served by msda_tiled6.hip (launched ahead)
    typedef ValueIO<BF16> IO;
    typedef typename IO::elem_t elem_t;
    constexpr int CPL = IO::CPL;
    constexpr int D = CPL * LPG;
    constexpr int G = 64 / LPG;                 // queries per wave (all for the SAME head)
    constexpr int QPB = G * MSDA_WAVES;         // consecutive queries per block

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LP = L * P;
    const int LPs = LP | 1;                      // odd stride in LDS -> conflict-free b64 broadcasts
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPG;                  // query within the wave
    const int sub = lane % LPG;                  // lane within the (query, head) pair
    float2_t *s_xy = reinterpret_cast<float2_t *>(smem) + (size_t)wave * G * LPs;
    float *s_w = reinterpret_cast<float *>(smem + (size_t)MSDA_WAVES * G * LPs * sizeof(float2_t)) +
                 (size_t)wave * G * LPs;

    // Work decomposition: chunk c = (query tile t = c / M, head m = c % M); a block owns QPB consecutive
    // queries of ONE head, a wave G consecutive queries: neighbouring queries sample neighbouring pixels, so the
    // wave's corner gathers keep hitting the same 128-byte lines (vector L1) instead of 8 unrelated head slices.
    // XCD-aware walk: XCD x (= blockIdx % 8) owns the contiguous chunk range [x*cpx, (x+1)*cpx).
    const int xcd = blockIdx.x & 7;
    const long cpx = (n_chunks + 7) >> 3;
    const int blocks_per_xcd = gridDim.x >> 3;   // gridDim.x is a multiple of 8 (host guarantees)
    const long MD = (long)M * D;

    for (long j = blockIdx.x >> 3; j < cpx; j += blocks_per_xcd) {
        const long chunk = (long)xcd * cpx + j;
        if (chunk >= n_chunks) break;            // block-uniform
        const int m = (int)(chunk % M);
        const long bq0 = (chunk / M) * QPB + (long)wave * G;   // first (b*Lq+q) of this wave

        // ---- stage loc / weights of the wave's G (query, m) pairs in LDS ----
        __syncthreads();                          // previous iteration's LDS reads are done
        for (int i = lane; i < G * LP; i += 64) {
            const int g = i / LP, pnt = i - g * LP;
            long bq = bq0 + g;
            bq = bq < n_bq ? bq : n_bq - 1;
            const long pr = bq * M + m;
            s_xy[g * LPs + pnt] = *reinterpret_cast<const float2_t *>(loc + (pr * LP + pnt) * 2);
            s_w[g * LPs + pnt] = attw[pr * LP + pnt];
        }
        __syncthreads();

        long bq = bq0 + grp;
        const bool live = bq < n_bq;
        bq = live ? bq : n_bq - 1;
        const long b = bq / Lq;
        float acc[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

        // ---- gather ----
        const float2_t *my_xy = s_xy + grp * LPs;
        const float *my_w = s_w + grp * LPs;
        const elem_t *vb = value + (b * (long)S) * MD + (long)m * D + sub * CPL;
        constexpr int np = PT;                    // points per level (compile time: full unroll, 4*PT loads in flight)
        (void)P;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l];
            const int W = (int)shapes[2 * l + 1];
            // an empty level: the reference accepts the point (h_im = -0.5 > -1) but every corner test fails, so it adds
            // nothing and touches no memory -- the clamps below would otherwise produce index -1
            if (H <= 0 || W <= 0) continue;
            const elem_t *vl = vb + (long)lsi[l] * MD;
#pragma unroll
            for (int p = 0; p < np; ++p) {
                const float2_t xy = my_xy[l * np + p];
                const float aw = my_w[l * np + p];
                const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
                const int hl = sp.h_low, wl = sp.w_low;
                const float lh = sp.h_im - (float)hl, lw = sp.w_im - (float)wl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                // a rejected point (incl. NaN / inf locations) contributes exactly nothing, as in the reference
                const float w1 = sp.ok ? hh * hw : 0.f, w2 = sp.ok ? hh * lw : 0.f;
                const float w3 = sp.ok ? lh * hw : 0.f, w4 = sp.ok ? lh * lw : 0.f;
                const bool k1 = sp.ok && hl >= 0 && wl >= 0;
                const bool k2 = sp.ok && hl >= 0 && wl + 1 <= W - 1;
                const bool k3 = sp.ok && hl + 1 <= H - 1 && wl >= 0;
                const bool k4 = sp.ok && hl + 1 <= H - 1 && wl + 1 <= W - 1;
                // clamped addresses: always in bounds, so all loads are unconditional and independent
                const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
                const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);
                float v1[CPL], v2[CPL], v3[CPL], v4[CPL];
                IO::load(vl + ((long)h0 * W + x0) * MD, v1);
                IO::load(vl + ((long)h0 * W + x1) * MD, v2);
                IO::load(vl + ((long)h1 * W + x0) * MD, v3);
                IO::load(vl + ((long)h1 * W + x1) * MD, v4);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    // selects (not multiplies by 0): a non-finite value at a clamped address must not leak
                    const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f;
                    const float a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;
                    const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;
                    acc[c] += val * aw;
                }
            }
        }
        if (live) IO::store(out + (bq * M + m) * D + sub * CPL, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Generic forward: any D / M / L / P, float or double.  One thread per output scalar (b,q,m,c); used for
// shapes the vector kernel does not cover (D*sizeof not a power-of-two multiple of 16 B) and for fp64.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ attw, int S, int M, int D, int L, int P,
    long n_out, long pairs_per_batch, T *__restrict__ out)
{
    const long MD = (long)M * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_out; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long pair = idx / D;
        const int m = (int)(pair % M);
        const long b = pair / pairs_per_batch;
        long wptr = pair * L * P;
        T col = (T)0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const T *vl = value + (b * (long)S + (long)lsi[l]) * MD + (long)m * D + c;
            for (int p = 0; p < P; ++p, ++wptr) {
                const SamplePoint<T> sp = sample_point<T>(loc[2 * wptr], loc[2 * wptr + 1], H, W);
                if (!sp.ok) continue;
                const int hl = sp.h_low, wl = sp.w_low;
                const T lh = sp.h_im - (T)hl, lw = sp.w_im - (T)wl;
                const T hh = (T)1 - lh, hw = (T)1 - lw;
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (hl >= 0 && wl >= 0) v1 = vl[((long)hl * W + wl) * MD];
                if (hl >= 0 && wl + 1 <= W - 1) v2 = vl[((long)hl * W + wl + 1) * MD];
                if (hl + 1 <= H - 1 && wl >= 0) v3 = vl[((long)(hl + 1) * W + wl) * MD];
                if (hl + 1 <= H - 1 && wl + 1 <= W - 1) v4 = vl[((long)(hl + 1) * W + wl + 1) * MD];
                const T val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
                col += val * attw[wptr];
            }
        }
        out[idx] = col;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Sample-index probe (parity tooling that ships with the library: it runs the SAME sample_point()).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msda_sample_index_kernel(
    const int64_t *__restrict__ shapes, const float *__restrict__ loc, int L, int P, long n_points,
    int32_t *__restrict__ h_low, int32_t *__restrict__ w_low, uint8_t *__restrict__ mask)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_points; i += (long)gridDim.x * blockDim.x) {
        const int l = (int)((i / P) % L);
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const SamplePoint<float> sp = sample_point<float>(loc[2 * i], loc[2 * i + 1], H, W);
        uint8_t mk = sp.ok ? 1 : 0;
        int hl = 0, wl = 0;
        if (sp.ok) {
            hl = sp.h_low; wl = sp.w_low;
            if (hl >= 0 && wl >= 0) mk |= 2;
            if (hl >= 0 && wl + 1 <= W - 1) mk |= 4;
            if (hl + 1 <= H - 1 && wl >= 0) mk |= 8;
            if (hl + 1 <= H - 1 && wl + 1 <= W - 1) mk |= 16;
        }
        h_low[i] = hl; w_low[i] = wl; mask[i] = mk;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward (generic): one thread per sampling point (b,q,m,l,p); serial loop over the D channels, atomics
// into grad_value (the same scatter the reference does), plain stores for grad_loc / grad_attw (each point
// has exactly one owner thread).  Follows ms_deform_im2col_cuda.cuh:87-161 (col2im bilinear).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ attw, const T *__restrict__ grad_out,
    int S, int M, int D, int L, int P, long n_points, long pairs_per_batch,
    T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_attw)
{
    const long MD = (long)M * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_points; i += (long)gridDim.x * blockDim.x) {
        const int l = (int)((i / P) % L);
        const long pair = i / ((long)L * P);
        const int m = (int)(pair % M);
        const long b = pair / pairs_per_batch;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const SamplePoint<T> sp = sample_point<T>(loc[2 * i], loc[2 * i + 1], H, W);
        if (!sp.ok) continue;  // grads stay at the caller's zero fill
        const int hl = sp.h_low, wl = sp.w_low;
        const T lh = sp.h_im - (T)hl, lw = sp.w_im - (T)wl;
        const T hh = (T)1 - lh, hw = (T)1 - lw;
        const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const T aw = attw[i];
        const long base = (b * (long)S + (long)lsi[l]) * MD + (long)m * D;
        const long o1 = base + ((long)hl * W + wl) * MD, o2 = o1 + MD;
        const long o3 = o1 + (long)W * MD, o4 = o3 + MD;
        const bool k1 = hl >= 0 && wl >= 0, k2 = hl >= 0 && wl + 1 <= W - 1;
        const bool k3 = hl + 1 <= H - 1 && wl >= 0, k4 = hl + 1 <= H - 1 && wl + 1 <= W - 1;
        const T *go = grad_out + pair * D;
        T g_aw = 0, g_x = 0, g_y = 0;
        for (int c = 0; c < D; ++c) {
            const T top = go[c];
            const T tgv = top * aw;
            T ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
            if (k1) { v1 = value[o1 + c]; ghw -= hw * v1; gww -= hh * v1; atomicAdd(grad_value + o1 + c, w1 * tgv); }
            if (k2) { v2 = value[o2 + c]; ghw -= lw * v2; gww += hh * v2; atomicAdd(grad_value + o2 + c, w2 * tgv); }
            if (k3) { v3 = value[o3 + c]; ghw += hw * v3; gww -= lh * v3; atomicAdd(grad_value + o3 + c, w3 * tgv); }
            if (k4) { v4 = value[o4 + c]; ghw += lw * v4; gww += lh * v4; atomicAdd(grad_value + o4 + c, w4 * tgv); }
            const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
            g_aw += top * val;
            g_x += (T)W * gww * tgv;
            g_y += (T)H * ghw * tgv;
        }
        grad_attw[i] = g_aw;
        grad_loc[2 * i] = g_x;
        grad_loc[2 * i + 1] = g_y;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward, vectorised (fp32, D/4 a power of two <= 64): same lane mapping as the forward gather kernel -- an
// LPG-lane group per (b,q,m) pair, 4 channels per lane.  grad_value: hardware fp32 atomic adds (global_atomic_add_f32)
// of 4 channels per corner per lane -- the same scatter the reference does with atomicAdd, 16 bytes wide per lane;
// grad_sampling_loc / grad_attn_weight: per-lane partial over its 4 channels, then a log2(LPG)-step shuffle reduction
// inside the group; one lane stores (each point has exactly one owner group, the outputs are plain stores).
// Follows ms_deform_im2col_cuda.cuh:87-161 (col2im bilinear) and :301-360.
// ---------------------------------------------------------------------------------------------------------
template <int LPG, int PT>
__global__ __launch_bounds__(MSDA_BLOCK) void msda_bwd_vec_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, const float *__restrict__ grad_out, int S, int M,
    int L, long n_pairs, long pairs_per_batch, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attw)
{
    constexpr int D = 4 * LPG;
    constexpr int G = 64 / LPG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPG, sub = lane % LPG;
    const long MD = (long)M * D;
    for (long base = ((long)blockIdx.x * MSDA_WAVES + wave) * G; base < n_pairs; base += (long)gridDim.x * MSDA_WAVES * G) {
        long pair = base + grp;
        const bool live = pair < n_pairs;
        pair = live ? pair : n_pairs - 1;
        const int m = (int)(pair % M);
        const long b = pair / pairs_per_batch;
        const float4_t go = *reinterpret_cast<const float4_t *>(grad_out + pair * D + sub * 4);
        const long vb = (b * (long)S) * MD + (long)m * D + sub * 4;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            if (H <= 0 || W <= 0) continue;   // empty level: no corner is inside, all gradients keep the caller's zero fill
            const long vl = vb + (long)lsi[l] * MD;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const long pi = (pair * L + l) * PT + p;
                const float2_t xy = *reinterpret_cast<const float2_t *>(loc + pi * 2);
                const float aw = attw[pi];
                const SamplePoint<float> sp = sample_point<float>(xy.x, xy.y, H, W);
                const bool pok = sp.ok && live;
                const int hl = sp.h_low, wl = sp.w_low;
                const float lh = sp.h_im - (float)hl, lw = sp.w_im - (float)wl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const bool k1 = pok && hl >= 0 && wl >= 0, k2 = pok && hl >= 0 && wl + 1 <= W - 1;
                const bool k3 = pok && hl + 1 <= H - 1 && wl >= 0, k4 = pok && hl + 1 <= H - 1 && wl + 1 <= W - 1;
                const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
                const int x0 = min(max(wl, 0), W - 1), x1 = min(max(wl + 1, 0), W - 1);
                const long o1 = vl + ((long)h0 * W + x0) * MD, o2 = vl + ((long)h0 * W + x1) * MD;
                const long o3 = vl + ((long)h1 * W + x0) * MD, o4 = vl + ((long)h1 * W + x1) * MD;
                const float4_t v1 = *reinterpret_cast<const float4_t *>(value + o1);
                const float4_t v2 = *reinterpret_cast<const float4_t *>(value + o2);
                const float4_t v3 = *reinterpret_cast<const float4_t *>(value + o3);
                const float4_t v4 = *reinterpret_cast<const float4_t *>(value + o4);
                float g_aw = 0.f, g_x = 0.f, g_y = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float top = go[c], tgv = top * aw;
                    const float a1 = k1 ? v1[c] : 0.f, a2 = k2 ? v2[c] : 0.f, a3 = k3 ? v3[c] : 0.f, a4 = k4 ? v4[c] : 0.f;
                    // grad_h_weight / grad_w_weight of the reference (corner terms vanish when the corner is outside)
                    const float ghw = -hw * a1 - lw * a2 + hw * a3 + lw * a4;
                    const float gww = -hh * a1 + hh * a2 - lh * a3 + lh * a4;
                    if (k1) unsafeAtomicAdd(grad_value + o1 + c, w1 * tgv);
                    if (k2) unsafeAtomicAdd(grad_value + o2 + c, w2 * tgv);
                    if (k3) unsafeAtomicAdd(grad_value + o3 + c, w3 * tgv);
                    if (k4) unsafeAtomicAdd(grad_value + o4 + c, w4 * tgv);
                    const float val = w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;
                    g_aw += top * val;
                    g_x += (float)W * gww * tgv;
                    g_y += (float)H * ghw * tgv;
                }
#pragma unroll
                for (int o = LPG >> 1; o > 0; o >>= 1) {
                    g_aw += __shfl_xor(g_aw, o);
                    g_x += __shfl_xor(g_x, o);
                    g_y += __shfl_xor(g_y, o);
                }
                if (sub == 0 && pok) {   // rejected points keep the caller's zero fill
                    grad_attw[pi] = g_aw;
                    grad_loc[2 * pi] = g_x;
                    grad_loc[2 * pi + 1] = g_y;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------
static int check_dims(int B, int S, int M, int D, int L, int Lq, int P)
{
    VLLM_REQUIRE(B >= 0 && Lq >= 0 && S >= 0, "msda: negative batch/query/key count");
    VLLM_REQUIRE(M > 0 && D > 0 && L > 0 && P > 0, "msda: num_heads, channels, levels, points must be positive");
    return VLLM_OK;
}

static int cu_count() { return device_cus(); }

template <bool BF16, int LPG>
static int launch_vec(const void *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *attw, int B, int S, int M, int L, int Lq, int P, void *out, hipStream_t st,
                      int skip_pyramid = 0)
{
    typedef typename ValueIO<BF16>::elem_t elem_t;
    constexpr int G = 64 / LPG;
    const long n_bq = (long)B * Lq;
    const long n_chunks = ((n_bq + G * MSDA_WAVES - 1) / (G * MSDA_WAVES)) * M;
    const int LPs = (L * P) | 1;
    const size_t lds = (size_t)MSDA_WAVES * G * LPs * (sizeof(float2_t) + sizeof(float));
    long want = (n_chunks + 7) / 8;                       // blocks per XCD if one chunk each
    const long cap = ((long)cu_count() / 8) * 8;          // CUs per XCD x 8 resident 256-thread blocks per CU
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    const dim3 grid((unsigned)(want * 8)), block(MSDA_BLOCK);
#define VLLM_MSDA_LAUNCH(PT)                                                                                   \
    VLLM_LAUNCH((msda_fwd_vec_kernel<BF16, LPG, PT>), grid, block, lds, st, (const elem_t *)value,      \
                       shapes, lsi, loc, attw, S, M, L, P, n_bq, (long)Lq, n_chunks, (elem_t *)out, skip_pyramid)
    if (P == 4) VLLM_MSDA_LAUNCH(4);
    else if (P == 8) VLLM_MSDA_LAUNCH(8);
    else if (P == 2) VLLM_MSDA_LAUNCH(2);
    else VLLM_MSDA_LAUNCH(1);                             // vec_ok() admits only P in {1,2,4,8}
#undef VLLM_MSDA_LAUNCH
    VLLM_CHECK_LAUNCH("msda_fwd_vec_kernel");
    return VLLM_OK;
}

// returns true if the vector path can take (D, CPL): D/CPL a power of two <= 64 and the LDS slice small enough
static bool vec_ok(int D, int CPL, int L, int P, const void *value, const void *out)
{
    if (D % CPL) return false;
    if (P != 1 && P != 2 && P != 4 && P != 8) return false;   // other point counts -> generic kernel
    const int lpg = D / CPL;
    if (lpg < 1 || lpg > 64 || (lpg & (lpg - 1))) return false;
    if (!aligned16(value) || !aligned16(out)) return false;
    const size_t lds = (size_t)MSDA_WAVES * (64 / lpg) * ((L * P) | 1) * 12;
    return lds <= 48 * 1024;
}

template <bool BF16>
static int dispatch_vec(int lpg, const void *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                        const float *attw, int B, int S, int M, int L, int Lq, int P, void *out, hipStream_t st,
                        int skip_pyramid = 0)
{
    switch (lpg) {
#define C(N) case N: return launch_vec<BF16, N>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, P, out, st, skip_pyramid)
        C(1); C(2); C(4); C(8); C(16); C(32); C(64);
#undef C
    }
    set_error("msda: unsupported lanes-per-pair %d", lpg);
    return VLLM_EINVAL;
}

template <typename T>
static int launch_generic_fwd(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                              const T *attw, int B, int S, int M, int D, int L, int Lq, int P, T *out,
                              hipStream_t st)
{
    const long n_out = (long)B * Lq * M * D;
    long blocks = (n_out + 255) / 256;
    const long cap = (long)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    VLLM_LAUNCH((msda_fwd_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, value, shapes, lsi,
                       loc, attw, S, M, D, L, P, n_out, (long)Lq * M, out);
    VLLM_CHECK_LAUNCH("msda_fwd_generic_kernel");
    return VLLM_OK;
}

template <typename T>
static int launch_generic_bwd(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                              const T *attw, const T *grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                              T *gv, T *gl, T *gw, hipStream_t st)
{
    const long n_points = (long)B * Lq * M * L * P;
    if (n_points == 0) return VLLM_OK;
    long blocks = (n_points + 255) / 256;
    const long cap = (long)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    VLLM_LAUNCH((msda_bwd_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, value, shapes, lsi,
                       loc, attw, grad_out, S, M, D, L, P, n_points, (long)Lq * M, gv, gl, gw);
    VLLM_CHECK_LAUNCH("msda_bwd_generic_kernel");
    return VLLM_OK;
}

}  // namespace vllm

using namespace vllm;

namespace vllm {
// The fp32 operator for a caller that consumes the result in bf16 (msda_layer.hip).  *where = 0: result in `out` (fp32);
// 1: in `out16` when the level maps form a 2x pyramid, in `out` otherwise (the caller's conversion pass tests the same
// device-side predicate, geometry_is_pyramid).
int msda_forward_f32_out16(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                           int B, int S, int M, int D, int L, int Lq, int P, float *out, uint16_t *out16, int *where,
                           hipStream_t st, int geometry)
{
    *where = 0;
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    if (msda_tiled_ok(D, L, P, Lq, S, B, M, value, out, loc) && (reinterpret_cast<uintptr_t>(out16) & 7u) == 0)
        return msda_tiled_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, P, out, st, out16, where, geometry);
    return vllm_msda_forward_f32(value, shapes, lsi, loc, attw, B, S, M, D, L, Lq, P, out, (vllm_stream_t)st);
}
}  // namespace vllm

extern "C" int vllm_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                     const float *loc, const float *attw, int B, int S, int M, int D, int L,
                                     int Lq, int P, float *out, vllm_stream_t stream)
{
    return vllm_msda_forward_f32_geo(value, shapes, lsi, loc, attw, B, S, M, D, L, Lq, P, VLLM_GEO_UNKNOWN, out, stream);
}

extern "C" int vllm_msda_forward_f32_geo(const float *value, const int64_t *shapes, const int64_t *lsi,
                                         const float *loc, const float *attw, int B, int S, int M, int D, int L,
                                         int Lq, int P, int geometry, float *out, vllm_stream_t stream)
{
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    VLLM_REQUIRE(geometry == VLLM_GEO_UNKNOWN || geometry == VLLM_GEO_PYRAMID || geometry == VLLM_GEO_GENERAL || geometry == VLLM_GEO_NESTED,
                 "msda_forward_f32: geometry must be VLLM_GEO_UNKNOWN / _PYRAMID / _GENERAL / _NESTED (got %d)", geometry);
    if ((long)B * Lq == 0) return VLLM_OK;
    VLLM_REQUIRE(value && shapes && lsi && loc && attw && out, "msda_forward_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (msda_tiled_ok(D, L, P, Lq, S, B, M, value, out, loc)) {   // encoder self-attention shape: LDS-tiled kernel
        prof_mark(PT_MSDA_ENC, st);
        rc = msda_tiled_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, P, out, st, nullptr, nullptr, geometry);
    } else {
        prof_mark(PT_MSDA_OTHER, st);
        if (vec_ok(D, 4, L, P, value, out) && (reinterpret_cast<uintptr_t>(loc) & 7u) == 0)
            rc = dispatch_vec<false>(D / 4, value, shapes, lsi, loc, attw, B, S, M, L, Lq, P, out, st);
        else
            rc = launch_generic_fwd<float>(value, shapes, lsi, loc, attw, B, S, M, D, L, Lq, P, out, st);
    }
    prof_mark(PT_END, st);
    return rc;
}

extern "C" int vllm_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                                     const double *loc, const double *attw, int B, int S, int M, int D, int L,
                                     int Lq, int P, double *out, vllm_stream_t stream)
{
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    if ((long)B * Lq == 0) return VLLM_OK;
    VLLM_REQUIRE(value && shapes && lsi && loc && attw && out, "msda_forward_f64: null pointer");
    return launch_generic_fwd<double>(value, shapes, lsi, loc, attw, B, S, M, D, L, Lq, P, out, (hipStream_t)stream);
}

extern "C" int vllm_msda_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                                      const float *loc, const float *attw, int B, int S, int M, int D, int L,
                                      int Lq, int P, uint16_t *out, vllm_stream_t stream)
{
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    if ((long)B * Lq == 0) return VLLM_OK;
    VLLM_REQUIRE(value && shapes && lsi && loc && attw && out, "msda_forward_bf16: null pointer");
    VLLM_REQUIRE(vec_ok(D, 8, L, P, value, out) && (reinterpret_cast<uintptr_t>(loc) & 7u) == 0,
                 "msda_forward_bf16: needs D in {8,16,...,512} (D/8 a power of two) and 16-byte aligned tensors (D=%d)", D);
    hipStream_t st = (hipStream_t)stream;
    // encoder self-attention shape on a pyramid: the LDS-tiled kernel (value converted to fp32 while it is staged); it
    // returns at once for any other geometry and the gather kernel behind it then does the work
    const bool t6 = msda_tiled_enabled() == 1 && msda_tiled6_ok(D, L, P, Lq, S, B, M) && aligned16(loc) && aligned16(attw);
    if (t6)
        if (int e = msda_tiled6_launch_bf16(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st)) return e;
    return dispatch_vec<true>(D / 8, value, shapes, lsi, loc, attw, B, S, M, L, Lq, P, out, st, t6 ? 1 : 0);
}

extern "C" int vllm_msda_sample_index_f32(const int64_t *shapes, const float *loc, int B, int M, int L, int Lq,
                                          int P, int32_t *h_low, int32_t *w_low, uint8_t *mask,
                                          vllm_stream_t stream)
{
    VLLM_REQUIRE(B >= 0 && Lq >= 0 && M > 0 && L > 0 && P > 0, "msda_sample_index: bad dims");
    const long n = (long)B * Lq * M * L * P;
    if (n == 0) return VLLM_OK;
    VLLM_REQUIRE(shapes && loc && h_low && w_low && mask, "msda_sample_index: null pointer");
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    VLLM_LAUNCH(msda_sample_index_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, shapes,
                       loc, L, P, n, h_low, w_low, mask);
    VLLM_CHECK_LAUNCH("msda_sample_index_kernel");
    return VLLM_OK;
}

namespace vllm {
int msda_bwd_mfma_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                         const float *grad_out, int B, int S, int M, int L, int Lq, float *gv, float *gl, float *gw,
                         hipStream_t st);   // msda_bwd_mfma.hip
}  // namespace vllm

template <int LPG>
static int launch_bwd_vec(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *attw, const float *grad_out, int B, int S, int M, int L, int Lq, int P, float *gv,
                          float *gl, float *gw, hipStream_t st)
{
    constexpr int G = 64 / LPG;
    const long n_pairs = (long)B * Lq * M;
    long blocks = (n_pairs + G * MSDA_WAVES - 1) / (G * MSDA_WAVES);
    const long cap = (long)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks), block(MSDA_BLOCK);
#define LB(PT) VLLM_LAUNCH((msda_bwd_vec_kernel<LPG, PT>), grid, block, 0, st, value, shapes, lsi, loc, attw, grad_out, S, \
                           M, L, n_pairs, (long)Lq * M, gv, gl, gw)
    if (P == 4) LB(4); else if (P == 8) LB(8); else if (P == 2) LB(2); else LB(1);
#undef LB
    VLLM_CHECK_LAUNCH("msda_bwd_vec_kernel");
    return VLLM_OK;
}

// the matrix-core backward (msda_bwd_mfma.hip) takes the call: encoder self-attention shape, aligned operands, a (batch, head) slice
// addressed with 32-bit byte offsets, grad_sampling_loc stored in 8-byte pieces, a 32-bit item index
static bool bwd_takes_mfma(const float *value, const float *loc, const float *grad_out, const float *gv, const float *gl, int B, int S, int M,
                           int D, int L, int Lq, int P)
{
    return (long)B * Lq != 0 && msda_tiled_enabled() && D == 32 && P == 4 && L <= 8 /* BT_MAXL of msda_bwd_mfma.hip */ && Lq == S && Lq >= 4096 && aligned16(value) &&
           aligned16(grad_out) && aligned16(loc) && aligned16(gv) && (reinterpret_cast<uintptr_t>(gl) & 7u) == 0 &&
           (long)S * M * D * 4 < (1L << 31) && (long)B * M * Lq < (1L << 31);
}

extern "C" int vllm_msda_backward_f32_writes_point_grads(const float *value, const float *loc, const float *grad_out, const float *grad_value,
                                                         const float *grad_loc, int B, int S, int M, int D, int L, int Lq, int P)
{
    return bwd_takes_mfma(value, loc, grad_out, grad_value, grad_loc, B, S, M, D, L, Lq, P) ? 1 : 0;
}

extern "C" int vllm_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                      const float *loc, const float *attw, const float *grad_out, int B, int S,
                                      int M, int D, int L, int Lq, int P, float *gv, float *gl, float *gw,
                                      vllm_stream_t stream)
{
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    VLLM_REQUIRE((long)B * Lq == 0 || (value && shapes && lsi && loc && attw && grad_out && gv && gl && gw),
                 "msda_backward_f32: null pointer");
    // encoder self-attention shape: grad_value per (query tile, level) window as S^T x grad_out on the matrix cores (msda_bwd_mfma.hip;
    // the round-2 kernel that accumulated the window with LDS atomics is tools/experiments/msda_bwd_tiled.hip since round 5)
    if (bwd_takes_mfma(value, loc, grad_out, gv, gl, B, S, M, D, L, Lq, P))
        return msda_bwd_mfma_launch(value, shapes, lsi, loc, attw, grad_out, B, S, M, L, Lq, gv, gl, gw, (hipStream_t)stream);
    if ((long)B * Lq != 0 && D % 4 == 0 && (P == 1 || P == 2 || P == 4 || P == 8) && aligned16(value) &&
        aligned16(grad_out) && (reinterpret_cast<uintptr_t>(loc) & 7u) == 0) {
        const int lpg = D / 4;
        hipStream_t st = (hipStream_t)stream;
        switch (lpg) {
#define C(N) case N: return launch_bwd_vec<N>(value, shapes, lsi, loc, attw, grad_out, B, S, M, L, Lq, P, gv, gl, gw, st)
            C(1); C(2); C(4); C(8); C(16); C(32); C(64);
#undef C
        default: break;
        }
    }
    return launch_generic_bwd<float>(value, shapes, lsi, loc, attw, grad_out, B, S, M, D, L, Lq, P, gv, gl, gw,
                                     (hipStream_t)stream);
}

extern "C" int vllm_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                                      const double *loc, const double *attw, const double *grad_out, int B, int S,
                                      int M, int D, int L, int Lq, int P, double *gv, double *gl, double *gw,
                                      vllm_stream_t stream)
{
    if (int e = check_dims(B, S, M, D, L, Lq, P)) return e;
    VLLM_REQUIRE((long)B * Lq == 0 || (value && shapes && lsi && loc && attw && grad_out && gv && gl && gw),
                 "msda_backward_f64: null pointer");
    return launch_generic_bwd<double>(value, shapes, lsi, loc, attw, grad_out, B, S, M, D, L, Lq, P, gv, gl, gw,
                                      (hipStream_t)stream);
}
