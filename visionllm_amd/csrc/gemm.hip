// bf16 GEMM for the ViT / projector linears:  Y[M,N] = epilogue( X[M,K] @ W[N,K]^T + bias )   (nn.Linear layout)
//
// Replaces the rocBLAS/cuBLAS calls behind F.linear in
//   InternAttention.qkv / proj       (VisionLLMv2/visionllmv2/model/internvit/modeling_intern_vit.py:112, 124, 128, 141)
//   InternMLP.fc1 / fc2 + GELU       (:172-178)
//   LayerScale + residual            (:206-208)  -> fused into the producing GEMM's epilogue
//   patch-embedding Conv2d (as GEMM) (:73-75, 85-89)  -> EPI_EMBED adds bias + position embedding and scatters
//                                                        rows past the CLS slot
//   vl_bridge Linear/GELU            (visionllmv2/model/modeling_visionllmv2.py:162-182)
//
// gfx950 design: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Both operands are K-contiguous ("B^T input"), staged with LDS-DMA (global_load_lds_dwordx4: no VGPR round trip)
// into a double-buffered, XOR-swizzled image: LDS is lane-linear, so the swizzle is applied to the per-lane
// SOURCE address and again on the ds_read_b128 side (16-byte chunk c of row r lives at chunk c ^ (r & 7)), which
// makes every 16-lane ds_read_b128 group hit 16 distinct bank slots.  The MFMA operands are swapped
// (A := W rows, B := X rows) so each lane ends up with 4 consecutive output features of one token -> packed
// 8-byte bf16 stores and vector loads of bias / layer-scale / residual.  Tile -> block mapping is XCD-aware:
// an XCD owns a fixed subset of W panels (they stay in its 4 MiB L2) and streams the X panels.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

// Stage one 128 x 64 bf16 operand tile: 16 segments of 8 rows, 4 segments per wave, one LDS-DMA per segment.
__device__ __forceinline__ void stage_tile(const uint16_t *__restrict__ src, int ld, int row0, int nrows, int k0,
                                           char *lds_tile, int wave, int lane, int skipP = 0)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int seg = wave * 4 + s;
        const int r = seg * 8 + (lane >> 3);
        const int cp = lane & 7;                 // chunk position in the LDS row
        const int c = cp ^ (r & 7);              // source chunk that must land there
        int grow = row0 + r;
        grow = grow < nrows ? grow : nrows - 1;  // clamp: rows past the edge are computed but never stored
        if (skipP > 0) grow += grow / skipP + 1;  // X is [n, 1+P, K] and the CLS row of every image is skipped
        const uint16_t *g = src + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(lds_tile + seg * 1024), 16, 0, 0);
    }
}

// value of lane (quad_perm CTRL) of the same quad: 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ float quad_xchg(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(const GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][X tile | W tile]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping ----
    int tile = blockIdx.x;
    int tm_idx, tn_idx;
    {
        const int xcd = tile & 7, s = tile >> 3;
        if ((a.nt & 7) == 0) {                   // XCD x owns W panels {x, x+8, ...}
            const int npx = a.nt >> 3;
            tn_idx = xcd + 8 * (s % npx);
            tm_idx = s / npx;
        } else {                                 // generic: XCD x owns X panels {x, x+8, ...}, all W panels
            tm_idx = xcd + 8 * (s / a.nt);
            tn_idx = s % a.nt;
        }
        if (tm_idx >= a.mt || tn_idx >= a.nt) return;   // padding blocks (grid rounded up)
    }
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;

    f32x4_t acc[4][4];   // [tn][tm]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    const uint16_t *Wsrc = a.W;
    if (EPI == EPI_MSDA && n0 >= a.nsplit) Wsrc = a.W2 - (size_t)a.nsplit * a.ldw;   // second weight matrix (tile-uniform)
    stage_tile(a.X, a.ldx, m0, a.M, 0, smem, wave, lane, a.xP);
    stage_tile(Wsrc, a.ldw, n0, a.N, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, kq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        char *cur = smem + (kt & 1) * 2 * TILE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
        if (kt + 1 < nk) {
            stage_tile(a.X, a.ldx, m0, a.M, (kt + 1) * BK, nxt, wave, lane, a.xP);
            stage_tile(Wsrc, a.ldw, n0, a.N, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const char *xs = cur, *ws = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rw = wn * 64 + t * 16 + fr;
                wf[t] = *reinterpret_cast<const bf16x8_t *>(ws + rw * 128 + (((ks * 4 + kq) ^ (rw & 7)) << 4));
                const int rx = wm * 64 + t * 16 + fr;
                xf[t] = *reinterpret_cast<const bf16x8_t *>(xs + rx * 128 + (((ks * 4 + kq) ^ (rx & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (EPI == EPI_F32 || EPI == EPI_MSDA) {
        // ---- fp32 epilogues go through LDS: the accumulator layout would store 64-byte pieces of 16 different rows per
        // instruction (and an fp32 tile is all output traffic: these GEMMs are HBM-bound); re-read row-wise, a wave stores
        // two full 512-byte rows per instruction.  The ring is free after the K loop's last barrier: 128 x 512 B = 64 KiB.
        // 16-byte chunk c of row r lives at chunk c ^ (r & 15): conflict-free on the write side (16 rows per ds_write_b128
        // group) and on the read side (32 chunks of one row).
        float *ct = reinterpret_cast<float *>(smem);
        const int c = lane & 31, n = n0 + c * 4;
        const bool nok = n < a.N;
        // EPI_MSDA (MSDeformAttn.forward ms_deform_attn.py:110-129 on the accumulator): per-lane constants of column n
        // (the arithmetic below is msda_prep_kernel's, operation for operation: the two launch structures give the same bits)
        const bool is_off = EPI == EPI_MSDA && n < a.nsplit;
        int lvl = 0;
        float Wl = 1.f, Hl = 1.f;
        if (EPI == EPI_MSDA && is_off) {   // offsets in (head, level, point, xy) order: two points of level lvl
            lvl = (n / (2 * a.mP)) % a.mL;
            Wl = (float)a.shapes[2 * lvl + 1];
            Hl = (float)a.shapes[2 * lvl];
        }
        // the reference points of this lane's 16 rows: issued before the transpose so their latency hides behind it
        float4_t rp[16];
        if (EPI == EPI_MSDA && is_off) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int m = m0 + p * 8 + wave * 2 + (lane >> 5);
                const float *r = a.ref + ((size_t)(m < a.M ? m : a.M - 1) * a.mL + lvl) * a.ref_dim;
                if (a.ref_dim == 2) { const float2_t t2 = *reinterpret_cast<const float2_t *>(r); rp[p] = (float4_t){t2.x, t2.y, 0.f, 0.f}; }
                else rp[p] = *reinterpret_cast<const float4_t *>(r);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + fr, chunk = wn * 16 + i * 4 + kq;
                *reinterpret_cast<f32x4_t *>(ct + row * 128 + ((chunk ^ (row & 15)) << 2)) = acc[i][j];
            }
        __syncthreads();
        const EpiCols cols = epi_cols<EPI>(a, nok ? n : 0);
        const float invW = 1.f / Wl, invH = 1.f / Hl;
        const uint8_t *mask = EPI == EPI_F32 ? reinterpret_cast<const uint8_t *>(a.res) : nullptr;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int row = p * 8 + wave * 2 + (lane >> 5);
            const int m = m0 + row;
            const bool live = nok && m < a.M;
            const int mc = m < a.M ? m : a.M - 1;
            const f32x4_t t = *reinterpret_cast<const f32x4_t *>(ct + row * 128 + ((c ^ (row & 15)) << 2));
            float v[4] = {t[0] + cols.bia[0], t[1] + cols.bia[1], t[2] + cols.bia[2], t[3] + cols.bia[3]};
            if (EPI == EPI_F32) {
                const bool dead = mask && mask[mc] != 0;   // key-padding mask: zero rows (ms_deform_attn.py:106-109)
                const float4_t o4 = {dead ? 0.f : v[0], dead ? 0.f : v[1], dead ? 0.f : v[2], dead ? 0.f : v[3]};
                if (live) *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(a.Y) + (size_t)m * a.ldy + n) = o4;
            } else if (is_off) {
                const float r[4] = {rp[p][0], rp[p][1], rp[p][2], rp[p][3]};
                float sx, sy;
                if (a.ref_dim == 2) { sx = invW; sy = invH; }
                else if (a.four_d) { sx = r[2] * 0.5f / Wl; sy = r[3] * 0.5f / Hl; }
                else { sx = r[2] * 0.5f / (float)a.mP; sy = r[3] * 0.5f / (float)a.mP; }
                const float rx = r[0], ry = r[1];
                const float4_t o4 = {rx + v[0] * sx, ry + v[1] * sy, rx + v[2] * sx, ry + v[3] * sy};
                if (live) *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(a.Y) + (size_t)m * a.ldy + n) = o4;
            } else {   // the L * P == 16 logits of one head are the 4 lanes of a quad: softmax with two DPP exchanges
                float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                mx = fmaxf(mx, quad_xchg<0xB1>(mx));
                mx = fmaxf(mx, quad_xchg<0x4E>(mx));
                float e[4], sum = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { e[k] = __expf(v[k] - mx); sum += e[k]; }
                sum += quad_xchg<0xB1>(sum);
                sum += quad_xchg<0x4E>(sum);
                const float inv = 1.f / sum;
                const float4_t o4 = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
                if (live) *reinterpret_cast<float4_t *>(a.Y2 + (size_t)m * a.ldy2 + (n - a.nsplit)) = o4;
            }
        }
        return;
    }
    if (!a.direct_store) {
        // ---- bf16 epilogues, same route: row-wise from LDS a lane owns 8 consecutive features (one 16-byte store, 256
        // contiguous bytes per row, 16-byte residual / position loads) instead of 8 bytes in each of 16 rows ----
        float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + fr, chunk = wn * 16 + i * 4 + kq;
                *reinterpret_cast<f32x4_t *>(ct + row * 128 + ((chunk ^ (row & 15)) << 2)) = acc[i][j];
            }
        __syncthreads();
        const int c8 = lane & 15, n = n0 + c8 * 8;
        if (n >= a.N) return;   // (N % 8 == 0 on this route; no barrier follows)
        const EpiCols cols0 = epi_cols<EPI>(a, n), cols1 = epi_cols<EPI>(a, n + 4);
#pragma unroll 4
        for (int p = 0; p < 8; ++p) {
            const int row = p * 16 + wave * 4 + (lane >> 4);
            const int m = m0 + row;
            if (m >= a.M) continue;
            const f32x4_t t0 = *reinterpret_cast<const f32x4_t *>(ct + row * 128 + (((2 * c8) ^ (row & 15)) << 2));
            const f32x4_t t1 = *reinterpret_cast<const f32x4_t *>(ct + row * 128 + (((2 * c8 + 1) ^ (row & 15)) << 2));
            float v0[4], v1[4];
            epi_value<EPI>(a, m, n, t0, cols0, v0);
            epi_value<EPI>(a, m, n + 4, t1, cols1, v1);
            uint4_t o;
            o.x = pack_bf16x2(v0[0], v0[1]); o.y = pack_bf16x2(v0[2], v0[3]);
            o.z = pack_bf16x2(v1[0], v1[1]); o.w = pack_bf16x2(v1[2], v1[3]);
            *reinterpret_cast<uint4_t *>(a.Y + epi_out_row<EPI>(a, m) * a.ldy + n) = o;
        }
        return;
    }
    // ---- epilogue: lane holds features n..n+3 (rows of the swapped product) of token m ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + kq * 4;
        if (n >= a.N) continue;
        const EpiCols cols = epi_cols<EPI>(a, n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + fr;
            if (m >= a.M) continue;
            epi_store<EPI>(a, m, n, acc[i][j], cols);
        }
    }
}

int gemm256_bf16_launch(int epi, GemmArgs a, hipStream_t st);   // gemm256.hip

int gemm_bf16_launch(int epi, GemmArgs a, hipStream_t st)
{
    VLLM_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0, "gemm: bad dims M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.M == 0) return VLLM_OK;
    VLLM_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    VLLM_REQUIRE(a.N % 4 == 0, "gemm: N=%d must be a multiple of 4", a.N);
    VLLM_REQUIRE(a.ldx % 8 == 0 && a.ldw % 8 == 0 && a.ldy % 4 == 0 && aligned16(a.X) && aligned16(a.W) &&
                     (reinterpret_cast<uintptr_t>(a.Y) & (epi == EPI_F32 || epi == EPI_MSDA ? 15u : 7u)) == 0,
                 "gemm: operands must be 16-byte aligned with row strides multiple of 8 elements");
    VLLM_REQUIRE(epi != EPI_RESIDUAL || (a.res && a.ldr % 4 == 0), "gemm: residual epilogue needs res");
    VLLM_REQUIRE(epi != EPI_EMBED || (a.res && a.P > 0), "gemm: embed epilogue needs the position table and P");
    // tall, skinny K = 256 problems (the linears of a deformable-attention layer): weight-stationary streaming kernel, bit-identical
    // to the 128 x 128 kernel below where both serve the shape (its EPI_MSDA form also takes L = 1 ... 3 levels, which the tile
    // kernel's epilogue does not).  A forced variant (tests / tuning) keeps its kernel.
    if (a.variant == 0 && gemm_skinny_takes(epi, a)) return gemm_skinny_launch(epi, a, st);
    if (epi == EPI_MSDA) {
        VLLM_REQUIRE(a.W2 && a.Y2 && a.ref && a.shapes && a.mL > 0 && a.mP > 0 && a.mP % 2 == 0 && a.mL * a.mP == 16 &&
                         a.nsplit % BN == 0 && a.nsplit > 0 && a.nsplit < a.N && (a.N - a.nsplit) % 16 == 0 && a.ldy2 % 4 == 0 &&
                         aligned16(a.W2) && aligned16(a.Y2) && aligned16(a.Y) && (a.ref_dim == 2 || a.ref_dim == 4) &&
                         (reinterpret_cast<uintptr_t>(a.ref) & (a.ref_dim == 4 ? 15u : 7u)) == 0,
                     "gemm: bad operands for the MSDA sampling epilogue");
    }
    if (epi == EPI_MSDA) a.variant = 1;
    if (a.variant == 4) { a.variant = 2; a.variant256 = 5; }   // 8-phase schedule on the 32x32x16 instruction
    VLLM_REQUIRE(a.variant != 3, "gemm: the 4-wave 128x128-per-wave variant lives in tools/experiments (not built)");
    if (a.ln_in || a.ln_out) {   // folded norm: only the 8-phase kernel implements it
        VLLM_REQUIRE(a.variant != 1 && epi != EPI_MSDA, "gemm: a folded norm needs the 8-phase kernel");
        return gemm256_bf16_launch(epi, a, st);
    }
    if (a.variant != 1 && (a.variant == 2 || (a.N >= 1024 && a.M >= 1024)))
        return gemm256_bf16_launch(epi, a, st);
    // bf16 epilogues of this kernel: through LDS (0) when 16-byte rows are possible, else straight from the accumulators (1)
    if (epi != EPI_F32 && epi != EPI_MSDA) {
        const bool rows16 = a.N % 8 == 0 && a.ldy % 8 == 0 && aligned16(a.Y);
        if (a.direct_store == 2 || a.direct_store == 0) a.direct_store = rows16 ? 0 : 1;
    }
    a.mt = ceil_div(a.M, BM);
    a.nt = ceil_div(a.N, BN);
    long tiles;
    if ((a.nt & 7) == 0) tiles = (long)a.mt * a.nt;
    else tiles = (long)((a.mt + 7) / 8) * 8 * a.nt;   // generic mapping pads the X-panel count to 8
    const dim3 grid((unsigned)tiles), block(GEMM_THREADS);
    const size_t lds = 4 * TILE_BYTES;
#define L(E) VLLM_LAUNCH((gemm_bf16_kernel<E>), grid, block, lds, st, a)
    switch (epi) {
    case EPI_BIAS: L(EPI_BIAS); break;
    case EPI_GELU: L(EPI_GELU); break;
    case EPI_QUICK_GELU: L(EPI_QUICK_GELU); break;
    case EPI_RESIDUAL: L(EPI_RESIDUAL); break;
    case EPI_EMBED: L(EPI_EMBED); break;
    case EPI_F32: L(EPI_F32); break;
    case EPI_MSDA: L(EPI_MSDA); break;
    default: set_error("gemm: unknown epilogue %d", epi); return VLLM_EINVAL;
    }
#undef L
    VLLM_CHECK_LAUNCH("gemm_bf16_kernel");
    return VLLM_OK;
}

}  // namespace vllm

using namespace vllm;

extern "C" int vllm_gemm_bf16(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y, int M, int N,
                              int K, int ldx, int ldw, int ldy, int epilogue, const uint16_t *scale,
                              const uint16_t *res, int ldr, int P, vllm_stream_t stream)
{
    VLLM_REQUIRE(X && W && Y, "vllm_gemm_bf16: null pointer");
    GemmArgs a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.scale = scale; a.res = res;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.P = P; a.mt = a.nt = 0; a.xP = 0; a.variant = gemm_variant_override(); a.variant256 = 0; a.direct_store = gemm_direct_store();
    if ((epilogue >> 8) & 3) a.variant = (epilogue >> 8) & 3;   // VLLM_GEMM_FORCE_* (tests / tuning)
    if (epilogue & 0x800) a.variant = 4;                         // VLLM_GEMM_FORCE_MF32
    if (epilogue & 0x1000) a.no_persist = 1;                     // VLLM_GEMM_FORCE_TILEWISE
    if (((epilogue >> 8) & 3) == 3) { a.variant = 2; a.variant256 = 3; }   // VLLM_GEMM_FORCE_192
    else if (((epilogue >> 8) & 3) == 2) a.variant256 = 4;
    return gemm_bf16_launch(epilogue & 0xff, a, (hipStream_t)stream);
}

extern "C" int vllm_gemm_bf16_ln(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y, int M, int N, int K, int ldx,
                                 int ldw, int ldy, int epilogue, const uint16_t *scale, const uint16_t *res, int ldr, float *ln_out,
                                 const float *ln_in, int ln_slots, int ln_rms, float ln_eps, const float *ln_colsum,
                                 const float *ln_bias, vllm_stream_t stream)
{
    VLLM_REQUIRE(X && W && Y, "vllm_gemm_bf16_ln: null pointer");
    GemmArgs a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.scale = scale; a.res = res;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.P = 0; a.mt = a.nt = 0; a.xP = 0; a.variant = 0; a.variant256 = 0; a.direct_store = gemm_direct_store();
    if (((epilogue >> 8) & 3) == 3) { a.variant = 2; a.variant256 = 3; }
    else if (((epilogue >> 8) & 3) == 2) { a.variant = 2; a.variant256 = 4; }
    if (epilogue & 0x1000) a.no_persist = 1;
    a.ln_out = ln_out; a.ln_in = ln_in; a.ln_slots = ln_slots; a.ln_cols = K; a.ln_rms = ln_rms; a.ln_eps = ln_eps;
    a.ln_colsum = ln_colsum; a.ln_bias = ln_bias;
    // round 5: RMSNorm rows of a hidden size other than four column tiles use the WIDE statistics layout ([M][16] floats, header)
    a.ln_wide = (ln_rms && ln_slots != 4 && ln_slots >= 1 && ln_slots <= 16) ? 1 : 0;
    return gemm_bf16_launch(epilogue & 0xff, a, (hipStream_t)stream);
}

extern "C" long vllm_gemm_scratch_bytes(void) { return SK_SCRATCH_BYTES; }
namespace vllm { long gemm256_sk_launches(); }
extern "C" long vllm_gemm_sk_launches(void) { return vllm::gemm256_sk_launches(); }
extern "C" long vllm_gemm_half_tail_launches(void) { return vllm::gemm256p_half_launches(); }
namespace vllm { long gemm256p_launches(); }
extern "C" long vllm_gemm_persistent_launches(void) { return vllm::gemm256p_launches(); }

extern "C" int vllm_gemm_bf16_sk(const uint16_t *X, const uint16_t *W, const uint16_t *bias, uint16_t *Y, int M, int N,
                                 int K, int ldx, int ldw, int ldy, int epilogue, const uint16_t *scale,
                                 const uint16_t *res, int ldr, int P, void *scratch, long scratch_bytes, vllm_stream_t stream)
{
    VLLM_REQUIRE(X && W && Y, "vllm_gemm_bf16_sk: null pointer");
    VLLM_REQUIRE(!scratch || (scratch_bytes >= SK_FLAG_BYTES + SK_SLOT_BYTES && aligned16(scratch)), "vllm_gemm_bf16_sk: scratch too small or misaligned");
    GemmArgs a;
    gemm_set_scratch(a, scratch, scratch_bytes);
    // the flags are reset in front of every call of THIS entry (a memset node of 4 KB: a call that died half way, or a caller that
    // never zeroed its scratch, must not hand a stale "slot ready" to the next one); the orchestrators reset theirs once per forward
    if (scratch) VLLM_REQUIRE(hipMemsetAsync(scratch, 0, SK_FLAG_BYTES, (hipStream_t)stream) == hipSuccess, "vllm_gemm_bf16_sk: flag reset failed");
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.scale = scale; a.res = res;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldr = ldr; a.P = P; a.mt = a.nt = 0; a.xP = 0; a.variant = gemm_variant_override(); a.variant256 = 0; a.direct_store = gemm_direct_store();
    if ((epilogue >> 8) & 3) a.variant = (epilogue >> 8) & 3;
    if (epilogue & 0x800) a.variant = 4;
    if (epilogue & 0x1000) a.no_persist = 1;
    if (((epilogue >> 8) & 3) == 3) { a.variant = 2; a.variant256 = 3; }
    else if (((epilogue >> 8) & 3) == 2) a.variant256 = 4;
    return gemm_bf16_launch(epilogue & 0xff, a, (hipStream_t)stream);
}
