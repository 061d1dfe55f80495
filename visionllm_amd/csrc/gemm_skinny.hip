// bf16 GEMM for TALL, SKINNY problems with K = 256: the three linears of a deformable-attention layer at d_model = 256
// (value_proj: M = B * S rows -> 256 fp32 features; sampling_offsets | attention_weights: M = B * Lq rows -> 256 + 128 features
// with the softmax / location epilogue; output_proj: 256 bf16 features).  Reference: MSDeformAttn.forward
// unipose/ops/modules/ms_deform_attn.py:104-111, 144; ...mask_dn.py:729-782.
//
// These GEMMs are HBM-bound by a wide margin (39 - 59 GFLOP against 0.3 - 0.6 GB per call at cfg 4, B = 8: the matrix pipe needs
// ~20 us, the memory system 40 - 80), and the 128 x 128 tile kernel they ran on is built for the opposite regime: 4 K tiles
// per block with a full drain (vmcnt(0) + barrier) behind each, the weight re-staged through LDS for every 128 rows, 0.34 - 0.57
// of the HBM roofline.  Here the roles are the ones the shape asks for:
//   * the WEIGHT IS STATIONARY IN REGISTERS: wave w of the 8 owns 32 (48) output features and holds their [features x 256]
//     slice as MFMA A fragments for the whole launch -- 64 (96) VGPRs, loaded once per block;
//   * the ACTIVATION ROWS STREAM through a 4-stage LDS ring of 64-row chunks (32 KiB each) filled by LDS-DMA
//     (buffer_load ... lds, 1 KiB = two rows per instruction, source-side XOR swizzle so that the B-fragment ds_read_b128 are
//     conflict-free; rows beyond M come back as zeros from the descriptor's range check): three chunks = 96 KiB per CU are in
//     flight or landed ahead of the one being multiplied, one counted vmcnt + ONE barrier per chunk;
//   * every wave multiplies its feature slice with ALL 64 rows of the chunk (the chunk is read from LDS once per wave: 256 KiB of
//     ds_read_b128 per chunk and CU, a fifth of the LDS pipe at HBM speed), fragments requested one step (4 reads) ahead with
//     inline ds_read and counted lgkmcnt -- a compiler-visible LDS read behind an LDS-DMA would be protected by vmcnt(0),
//     i.e. drain the ring;
//   * PERSISTENT: one block per CU walks chunks b, b + G, ...; the stores of chunk i (straight from the accumulators: a lane owns
//     4 consecutive features of one row, 16 B fp32 / 8 B bf16, the two 16-feature tiles of a wave complete 128-byte lines) drain
//     under the multiplication of chunk i + 1.
// Same MFMA (16x16x32 bf16, weight as the A operand), same K order and the same epilogue arithmetic as gemm_bf16_kernel: the
// results are BIT-IDENTICAL to the 128 x 128 kernel's (tests/test_msda_gpu.py::test_skinny_gemm_*), including EPI_MSDA's softmax
// (the four lanes that hold a head's 16 logits are 16 / 32 lanes apart here instead of neighbours: same reduction tree).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace vllm {

namespace {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

constexpr int SK_K = 256, SK_KS = SK_K / 32, SK_R = 64, SK_STAGE = SK_R * SK_K * 2, SK_S = 4, SK_THREADS = 512;
constexpr int SK_AUX = 8 * 1024;           // per stage: one 1 KiB slot per wave for the chunk's per-row epilogue inputs
constexpr int SK_LDS = SK_S * (SK_STAGE + SK_AUX);   // 160 KiB

// x[lane] <- f(x[lane], x[lane ^ 16]) / f(x[lane], x[lane ^ 32]) for a commutative f: v_permlane16_swap / v_permlane32_swap of a
// register with a copy of itself leave {own, partner} in the two results (which is which depends on the lane row; f does not
// care).  Inline with wait states in front and behind, as in gemm256p.hip (the builtin loses its second result on this toolchain).
__device__ __forceinline__ void xchg16(float x, float &p, float &q)
{
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    p = __builtin_bit_cast(float, a); q = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ void xchg32(float x, float &p, float &q)
{
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    p = __builtin_bit_cast(float, a); q = __builtin_bit_cast(float, b);
}

// One 16-byte output store; the data registers stay untouched for 16 wait states behind it (gemm256p.hip, property 1: a write to
// the first data register right behind a 16-byte buffer store with a scalar offset corrupts the stored dword).
__device__ __forceinline__ void sk_store16(u32x4_t o, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)voff, (int)soff, 0);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(o));
}

#define SK_READ(DST, ADDR, IMM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(IMM))

// AUX: the per-row epilogue inputs of a chunk travel WITH the chunk (one more LDS-DMA per wave and chunk, same counted wait) -- a
// register load inside the chunk loop would have to be waited for with vmcnt, and loads return in order: it would drain the ring.
//   0 none | 1 EPI_F32 / EPI_BIAS: the key-padding mask's 64 bytes (every wave its own copy: 16 lanes x 4 bytes)
//   2 / 3 EPI_MSDA: the chunk's reference points, 64 rows x L x {2, 4} floats = 2 / 4 KiB (waves 0-1 / 0-3 one KiB each, the
//   others a dummy that reads nothing)
template <int EPI, int NWC, int AUX>
__global__ __launch_bounds__(SK_THREADS, 1) void gemm_skinny_kernel(const GemmArgs a)
{
    static_assert((EPI == EPI_F32 && AUX <= 1 && NWC == 2) || (EPI == EPI_BIAS && AUX <= 1 && NWC == 2) || (EPI == EPI_MSDA && NWC == 3 && (AUX == 2 || AUX == 3)),
                  "skinny GEMM: fp32 (+ mask) / bias / MSDA epilogues");
    constexpr int NL = 4 + (AUX ? 1 : 0);   // VMEM loads per wave and chunk
    constexpr int RD = AUX == 3 ? 4 : 2;    // EPI_MSDA: floats per reference point
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();   // (LDS addresses below are absolute)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int fr = lane & 15, kq = lane >> 4;
    const int G = gridDim.x;
    const int nchunks = (a.M + SK_R - 1) / SK_R;
    const int n_it = (nchunks - (int)blockIdx.x + G - 1) / G;   // >= 1: the launcher never starts more blocks than chunks

    // ---- the wave's feature tiles (16 features each): NWC consecutive tiles.  EPI_MSDA: wave w = head w, for L <= 4 levels of P = 4
    // points: its 8 L offsets in two tiles of W (slots beyond 8 L are padding: computed from a repeated weight row, never
    // stored) and its 4 L logits in one tile of W2 (padding slots take no part in the softmax) ----
    const int mL = EPI == EPI_MSDA ? a.mL : 4, OFFH = 8 * mL, LGH = 4 * mL;
    int nt[NWC];
#pragma unroll
    for (int i = 0; i < NWC; ++i) nt[i] = (wave * NWC + i) * 16;
    bf16x8_t wf[NWC][SK_KS];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        const uint16_t *src;
        if (EPI == EPI_MSDA) {
            const int f = i * 16 + fr;
            src = i < 2 ? a.W + (size_t)(wave * OFFH + (f < OFFH ? f : OFFH - 1)) * a.ldw : a.W2 + (size_t)(wave * LGH + (fr < LGH ? fr : LGH - 1)) * a.ldw;
        } else src = a.W + (size_t)(nt[i] + fr) * a.ldw;
#pragma unroll
        for (int ks = 0; ks < SK_KS; ++ks) wf[i][ks] = *reinterpret_cast<const bf16x8_t *>(src + ks * 32 + kq * 8);
    }
    // per-lane epilogue constants: the lane's 4 features of tile i (EPI_MSDA: slot 16 i + 4 kq of the head's offsets, 4 kq of its
    // logits; `live`: not a padding slot), their bias
    bool live[NWC];
    int col[NWC];      // first of the lane's 4 output columns in Y (tiles 0 .. ) / Y2 (EPI_MSDA tile 2)
    float bia[NWC][4];
#pragma unroll
    for (int i = 0; i < NWC; ++i) {
        const uint16_t *bp = a.bias;
        live[i] = true;
        col[i] = nt[i] + kq * 4;
        if (EPI == EPI_MSDA) {
            if (i < 2) { live[i] = i * 16 + kq * 4 < OFFH; col[i] = wave * OFFH + i * 16 + kq * 4; }
            else { live[i] = kq * 4 < LGH; col[i] = wave * LGH + kq * 4; bp = a.bias2; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bia[i][r] = 0.f;
        if (bp && live[i]) {
            const uint2_t b = *reinterpret_cast<const uint2_t *>(bp + col[i]);
            bia[i][0] = bf16lo_to_f32(b.x); bia[i][1] = bf16hi_to_f32(b.x); bia[i][2] = bf16lo_to_f32(b.y); bia[i][3] = bf16hi_to_f32(b.y);
        }
    }
    // EPI_MSDA: the level of this lane's two points in offset tile i  ((head, level, point, xy) order, P = 4 points of 2 coordinates
    // per level = 8 features: level (16 i + 4 kq) / 8); W, H of the two levels
    float lW[2] = {1.f, 1.f}, lH[2] = {1.f, 1.f};
    int lvl[2] = {0, 0};
    if (EPI == EPI_MSDA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lvl[i] = (i * 16 + kq * 4) / 8;
            lvl[i] = lvl[i] < mL ? lvl[i] : mL - 1;
            lW[i] = (float)a.shapes[2 * lvl[i] + 1];
            lH[i] = (float)a.shapes[2 * lvl[i]];
        }
    }

    // ---- LDS-DMA of a chunk: wave w fills rows 8 w .. 8 w + 7 (4 instructions of two rows); lane l of an instruction lands at
    // row (l >> 5), physical 16-byte chunk (l & 31) and fetches source chunk (l & 31) ^ (row & 15) ----
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, (int)(((unsigned)(a.M - 1) * (unsigned)a.ldx + (unsigned)SK_K) * 2u), 0x00020000);
    unsigned xvo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = (wave * 4 + s) * 2 + (lane >> 5);
        xvo[s] = ((unsigned)(lane >> 5) * (unsigned)a.ldx + (unsigned)(((lane & 31) ^ (row & 15)) * 8)) * 2u;
    }
    // aux descriptor / per-lane offset (EPI_MSDA: the chunk's reference points are 64 rows x L x RD floats = ROWB bytes per row,
    // contiguous; wave w fetches KiB w of them while that starts inside the block)
    const int ROWB = mL * RD * 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        AUX == 1 ? (EPI == EPI_BIAS ? (void *)a.row_mask : (void *)a.res) : (void *)a.ref, 0,
        AUX == 1 ? a.M : AUX ? (int)((unsigned)a.M * (unsigned)ROWB) : 0, 0x00020000);
    const unsigned avo = AUX == 1 ? (lane < 16 ? (unsigned)lane * 4u : 0x80000000u)
                                  : (wave * 1024 < SK_R * ROWB ? (unsigned)(wave * 1024 + lane * 16) : 0x80000000u);
    auto issue_chunk = [&](int it) {   // chunk blockIdx.x + it * G into stage it % SK_S (chunks past the end: all zeros, no traffic)
        const unsigned chunk = (unsigned)((int)blockIdx.x + it * G);
        const unsigned row0 = chunk * (unsigned)SK_R + (unsigned)wave * 8u;
        char *dst = smem + (it % SK_S) * SK_STAGE + wave * 4096;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned so = (row0 + 2u * s) * (unsigned)a.ldx * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(dst + s * 1024), 16, (int)xvo[s], (int)so, 0, 0);
        }
        if (AUX == 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void *)(smem + SK_S * SK_STAGE + (it % SK_S) * SK_AUX + wave * 1024), 4,
                                                     (int)avo, (int)(chunk * (unsigned)SK_R), 0, 0);
        else if (AUX)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void *)(smem + SK_S * SK_STAGE + (it % SK_S) * SK_AUX + wave * 1024), 16,
                                                     (int)avo, (int)(chunk * (unsigned)(SK_R * ROWB)), 0, 0);
    };
    // B fragment (rows j * 16 + fr, K step ks): physical chunk ((ks << 2) | kq) ^ fr = (ks << 2) ^ (kq ^ fr); bit 4 of it is ks >> 2
    // (plain + 256 bytes), bits 2 - 3 mix with (ks & 3): four per-lane bases, everything else is an immediate
    unsigned xa4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xa4[q] = (unsigned)(fr * 512 + ((((q << 2) ^ (kq ^ fr)) & 15) << 4));

    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.Y, 0, (int)(((unsigned)(a.M - 1) * (unsigned)a.ldy + (unsigned)(EPI == EPI_MSDA ? a.nsplit : a.N)) * (EPI == EPI_BIAS ? 2u : 4u)), 0x00020000);   // (EPI_MSDA: nsplit = 8 heads x 8 L offsets)
    const __amdgpu_buffer_rsrc_t y2rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.Y2, 0, EPI == EPI_MSDA ? (int)(((unsigned)(a.M - 1) * (unsigned)a.ldy2 + (unsigned)(a.N - a.nsplit)) * 4u) : 0, 0x00020000);

    // the weight / bias / level loads above are waited for HERE, with a wait the compiler sees: left to itself it waits for them at
    // their first use -- inside the chunk loop, as s_waitcnt vmcnt(0) in every iteration, which drains the ring
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    // ---- prologue: four chunks requested, the first one landed ----
    issue_chunk(0); issue_chunk(1); issue_chunk(2); issue_chunk(3);
    if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    f32x4_t acc[NWC][4];   // [feature tile][row tile]
    for (int it = 0; it < n_it; ++it) {
        const int m0 = ((int)blockIdx.x + it * G) * SK_R;
        // ---- multiply: 8 steps (row tile j = t >> 1, K half t & 1) of 4 fragment reads + 4 NWC MFMAs, reads one step ahead ----
        unsigned xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[q] = xa4[q] + (unsigned)((it % SK_S) * SK_STAGE);
        bf16x8_t xf[2][4];
#define SK_READ_STEP(T, BUF)                                                  \
        do {                                                                  \
            constexpr int j_ = (T) >> 1, h_ = (T) & 1;                        \
            SK_READ(xf[BUF][0], xb[0], j_ * 8192 + h_ * 256);                 \
            SK_READ(xf[BUF][1], xb[1], j_ * 8192 + h_ * 256);                 \
            SK_READ(xf[BUF][2], xb[2], j_ * 8192 + h_ * 256);                 \
            SK_READ(xf[BUF][3], xb[3], j_ * 8192 + h_ * 256);                 \
        } while (0)
#define SK_MMA_STEP(T, BUF, CNT)                                                                                             \
        do {                                                                                                                 \
            constexpr int j_ = (T) >> 1, h_ = (T) & 1;                                                                       \
            asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(xf[BUF][0]), "+v"(xf[BUF][1]), "+v"(xf[BUF][2]), "+v"(xf[BUF][3])); \
            __builtin_amdgcn_s_setprio(1);                                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                    \
                _Pragma("unroll") for (int i = 0; i < NWC; ++i)                                                              \
                    acc[i][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][h_ * 4 + q], xf[BUF][q],                      \
                                                                         (h_ == 0 && q == 0) ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[i][j_], 0, 0, 0); \
            __builtin_amdgcn_s_setprio(0);                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        } while (0)
        SK_READ_STEP(0, 0);
        SK_READ_STEP(1, 1); SK_MMA_STEP(0, 0, 4);
        SK_READ_STEP(2, 0); SK_MMA_STEP(1, 1, 4);
        SK_READ_STEP(3, 1); SK_MMA_STEP(2, 0, 4);
        SK_READ_STEP(4, 0); SK_MMA_STEP(3, 1, 4);
        SK_READ_STEP(5, 1); SK_MMA_STEP(4, 0, 4);
        SK_READ_STEP(6, 0); SK_MMA_STEP(5, 1, 4);
        SK_READ_STEP(7, 1); SK_MMA_STEP(6, 0, 4);
        SK_MMA_STEP(7, 1, 0);
#undef SK_READ_STEP
#undef SK_MMA_STEP
        // ---- the next chunk has landed (newer: two chunks = 8 loads of this wave; outstanding stores only make the wait
        // longer); every wave is done with this chunk's stage: refill it with chunk it + 4 ----
        if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        // the chunk's per-row inputs: landed with the chunk (this wave's own DMA for the mask; the barrier at the end of the
        // previous iteration for the reference points), read before the stage is handed back
        float4_t rp[AUX >= 2 ? 2 : 1][4];
        unsigned dead[4] = {0u, 0u, 0u, 0u};
        if constexpr (AUX == 1) {
            const unsigned ad = (unsigned)(SK_S * SK_STAGE + (it % SK_S) * SK_AUX + wave * 1024 + fr);
            asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:16\n\tds_read_u8 %2, %4 offset:32\n\tds_read_u8 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(dead[0]), "=&v"(dead[1]), "=&v"(dead[2]), "=&v"(dead[3]) : "v"(ad) : "memory");
        } else if constexpr (AUX == 2) {
            float2_t t2[2][4];
            const unsigned ab = (unsigned)(SK_S * SK_STAGE + (it % SK_S) * SK_AUX + fr * ROWB);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(t2[i][j]) : "v"(ab + (unsigned)(j * 16 * ROWB + lvl[i] * 8)));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t2[0][0]), "+v"(t2[0][1]), "+v"(t2[0][2]), "+v"(t2[0][3]), "+v"(t2[1][0]), "+v"(t2[1][1]), "+v"(t2[1][2]), "+v"(t2[1][3]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) rp[i][j] = (float4_t){t2[i][j].x, t2[i][j].y, 0.f, 0.f};
        } else if constexpr (AUX == 3) {
            const unsigned ab = (unsigned)(SK_S * SK_STAGE + (it % SK_S) * SK_AUX + fr * ROWB);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(rp[i][j]) : "v"(ab + (unsigned)(j * 16 * ROWB + lvl[i] * 16)));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rp[0][0]), "+v"(rp[0][1]), "+v"(rp[0][2]), "+v"(rp[0][3]), "+v"(rp[1][0]), "+v"(rp[1][1]), "+v"(rp[1][2]), "+v"(rp[1][3]));
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_chunk(it + SK_S);
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue, straight from the accumulators: lane = features nt[i] + 4 kq .. + 3 of row m0 + 16 j + fr ----
        if (EPI == EPI_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = (unsigned)(m0 + j * 16) * (unsigned)a.ldy * 4u;
#pragma unroll
                for (int i = 0; i < NWC; ++i) {
                    const unsigned vo = ((unsigned)fr * (unsigned)a.ldy + (unsigned)(nt[i] + kq * 4)) * 4u;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = dead[j] ? 0.f : acc[i][j][r] + bia[i][r];
                    const u32x4_t o = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
                    sk_store16(o, yrs, vo, so);
                }
            }
        } else if (EPI == EPI_BIAS) {
            // bf16 rows: the wave's two feature tiles change lane rows (v_permlane16_swap) so that a lane holds 8 CONSECUTIVE features
            // -- lane row kq: tile kq & 1, features 8 (kq >> 1) .. + 7 -- and one instruction stores 64 contiguous bytes per row
            const unsigned vo = ((unsigned)fr * (unsigned)a.ldy + (unsigned)(nt[kq & 1] + (kq >> 1) * 8)) * 2u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = (unsigned)(m0 + j * 16) * (unsigned)a.ldy * 2u;
                unsigned p0x = pack_bf16x2(acc[0][j][0] + bia[0][0], acc[0][j][1] + bia[0][1]), p0y = pack_bf16x2(acc[0][j][2] + bia[0][2], acc[0][j][3] + bia[0][3]);
                unsigned p1x = pack_bf16x2(acc[1][j][0] + bia[1][0], acc[1][j][1] + bia[1][1]), p1y = pack_bf16x2(acc[1][j][2] + bia[1][2], acc[1][j][3] + bia[1][3]);
                if (AUX == 1 && dead[j]) { p0x = 0u; p0y = 0u; p1x = 0u; p1y = 0u; }   // key-padding mask: the row is stored as zeros (row fr in every lane row)
                asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(p0x), "+v"(p1x));
                asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(p0y), "+v"(p1y));
                const u32x4_t o = {p0x, p0y, p1x, p1y};
                sk_store16(o, yrs, vo, so);
            }
        } else {
            // MSDeformAttn.forward ms_deform_attn.py:110-129 on the accumulator; the arithmetic is gemm_bf16_kernel<EPI_MSDA>'s (and
            // msda_prep_kernel's), operation for operation
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = (unsigned)(m0 + j * 16) * (unsigned)a.ldy * 4u;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned vo = live[i] ? ((unsigned)fr * (unsigned)a.ldy + (unsigned)col[i]) * 4u : 0x80000000u;
                    const float v[4] = {acc[i][j][0] + bia[i][0], acc[i][j][1] + bia[i][1], acc[i][j][2] + bia[i][2], acc[i][j][3] + bia[i][3]};
                    const float4_t r = rp[i][j];
                    float sx, sy;
                    if (RD == 2) { sx = 1.f / lW[i]; sy = 1.f / lH[i]; }
                    else { sx = r[2] * 0.5f / (a.four_d ? lW[i] : (float)a.mP); sy = r[3] * 0.5f / (a.four_d ? lH[i] : (float)a.mP); }
                    const float o0 = r[0] + v[0] * sx, o1 = r[1] + v[1] * sy, o2 = r[0] + v[2] * sx, o3 = r[1] + v[3] * sy;
                    const u32x4_t o = {__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1), __builtin_bit_cast(unsigned, o2), __builtin_bit_cast(unsigned, o3)};
                    sk_store16(o, yrs, vo, so);
                }
                {   // the head's 16 logits of row fr: 4 in this lane, the others 16 / 32 / 48 lanes away
                    const float ninf = -__builtin_huge_valf();
                    const float v[4] = {live[2] ? acc[2][j][0] + bia[2][0] : ninf, live[2] ? acc[2][j][1] + bia[2][1] : ninf,
                                        live[2] ? acc[2][j][2] + bia[2][2] : ninf, live[2] ? acc[2][j][3] + bia[2][3] : ninf};   // (padding: exp -> 0)
                    float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), p, q;
                    xchg16(mx, p, q); mx = fmaxf(p, q);
                    xchg32(mx, p, q); mx = fmaxf(p, q);
                    float e[4], sum = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { e[k] = __expf(v[k] - mx); sum += e[k]; }
                    xchg16(sum, p, q); sum = p + q;
                    xchg32(sum, p, q); sum = p + q;
                    const float inv = 1.f / sum;
                    const float o0 = e[0] * inv, o1 = e[1] * inv, o2 = e[2] * inv, o3 = e[3] * inv;
                    const u32x4_t o = {__builtin_bit_cast(unsigned, o0), __builtin_bit_cast(unsigned, o1), __builtin_bit_cast(unsigned, o2), __builtin_bit_cast(unsigned, o3)};
                    const unsigned vo2 = live[2] ? ((unsigned)fr * (unsigned)a.ldy2 + (unsigned)col[2]) * 4u : 0x80000000u;
                    sk_store16(o, y2rs, vo2, (unsigned)(m0 + j * 16) * (unsigned)a.ldy2 * 4u);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

}  // namespace

static int g_skinny = -1;
int gemm_skinny_enabled()
{
    if (g_skinny < 0) {
        const char *e = getenv("VLLM_GEMM_SKINNY");
        g_skinny = e ? atoi(e) != 0 : 1;
    }
    return g_skinny;
}
int gemm_skinny_set(int v) { const int old = gemm_skinny_enabled(); g_skinny = v != 0; return old; }

// Does the shape belong here?  K = 256, N = 256 (fp32 / bias epilogues) or 8 heads x (8 L offsets + 4 L logits), L <= 4 levels of 4 points
// (EPI_MSDA: 256 + 128 features at L = 4, 192 + 96 for the 3-level pixel decoder, msdeformattn_pixel_decoder.py:57-58), plain rows.
bool gemm_skinny_takes(int epi, const GemmArgs &a)
{
    if (!gemm_skinny_enabled() || a.K != SK_K || a.M < 4096 || a.xP != 0 || a.ln_in || a.ln_out) return false;
    if (a.ldx < SK_K || (long)a.M * a.ldx * 2 >= (1L << 31) || (long)a.M * a.ldy * 4 >= (1L << 31)) return false;
    if (epi == EPI_BIAS)   // (16-byte output stores; a row mask travels as dwords like EPI_F32's)
        return a.N == 256 && a.ldy >= 256 && a.ldy % 8 == 0 && aligned16(a.Y) &&
               (!a.row_mask || (a.M % 4 == 0 && (reinterpret_cast<uintptr_t>(a.row_mask) & 3u) == 0));
    if (epi == EPI_F32)   // (the mask travels as dwords: M % 4 == 0 keeps the last one inside the array)
        return a.N == 256 && a.ldy >= 256 && (!a.res || (a.M % 4 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 3u) == 0));
    if (epi == EPI_MSDA)   // 8 heads x L <= 4 levels x 4 points: N = 8 (8 L + 4 L)
        return a.mL >= 1 && a.mL <= 4 && a.mP == 4 && a.N == 96 * a.mL && a.nsplit == 64 * a.mL && a.ldy >= a.nsplit && a.ldy % 4 == 0 &&
               a.ldy2 >= a.N - a.nsplit && a.ldy2 % 4 == 0 && a.W2 && a.Y2 && a.ref && a.shapes && aligned16(a.ref) && aligned16(a.W2) &&
               aligned16(a.Y2) && (a.ref_dim == 2 || a.ref_dim == 4) && (long)a.M * a.ldy2 * 4 < (1L << 31);
    return false;
}

int gemm_skinny_launch(int epi, const GemmArgs &a, hipStream_t st)
{
    const int nchunks = (a.M + SK_R - 1) / SK_R;
    const int cus = device_cus();
    const dim3 grid((unsigned)(nchunks < cus ? nchunks : cus)), block(SK_THREADS);
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
#define SK_ATTR(...) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_skinny_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS)
        SK_ATTR(EPI_F32, 2, 0); SK_ATTR(EPI_F32, 2, 1); SK_ATTR(EPI_BIAS, 2, 0); SK_ATTR(EPI_BIAS, 2, 1); SK_ATTR(EPI_MSDA, 3, 2); SK_ATTR(EPI_MSDA, 3, 3);
#undef SK_ATTR
    }
    switch (epi) {
    case EPI_F32:
        if (a.res) VLLM_LAUNCH((gemm_skinny_kernel<EPI_F32, 2, 1>), grid, block, SK_LDS, st, a);
        else VLLM_LAUNCH((gemm_skinny_kernel<EPI_F32, 2, 0>), grid, block, SK_LDS, st, a);
        break;
    case EPI_BIAS:
        if (a.row_mask) VLLM_LAUNCH((gemm_skinny_kernel<EPI_BIAS, 2, 1>), grid, block, SK_LDS, st, a);
        else VLLM_LAUNCH((gemm_skinny_kernel<EPI_BIAS, 2, 0>), grid, block, SK_LDS, st, a);
        break;
    case EPI_MSDA:
        if (a.ref_dim == 2) VLLM_LAUNCH((gemm_skinny_kernel<EPI_MSDA, 3, 2>), grid, block, SK_LDS, st, a);
        else VLLM_LAUNCH((gemm_skinny_kernel<EPI_MSDA, 3, 3>), grid, block, SK_LDS, st, a);
        break;
    default: set_error("gemm_skinny: unknown epilogue %d", epi); return VLLM_EINVAL;
    }
    VLLM_CHECK_LAUNCH("gemm_skinny_kernel");
    return VLLM_OK;
}

}  // namespace vllm
