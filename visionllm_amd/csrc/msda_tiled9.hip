// Multi-scale deformable attention forward, LDS-tiled kernel, generation 9 ("msda_tiled" option 20; the automatic choice since the end
// of round 4): generation 8's pyramid items and two teams half a period apart with a software-pipelined gather and straight-line
// halves.  LIBRARY BUILD: twelve waves (teams of six, T9_TW = 6 -- generation 8's occupancy), the gathering wave at s_setprio 1, and the
// preparing team gathering level 0 of pass 0 of its own item at the end of its half (T9_EARLY = 1): 439.6 us against generation 8's
// 454.7 on the same box (profiles/r04_msda9_two_per_simd.txt).  The text below describes the EIGHT-wave form the file started as
// (-DT9_TW=4: 508 us); both build from this file (tools/msda9_variants.sh).
//
// What generation 8's clocks showed (profiles/r03_msda8_two_teams.txt): a wave of the gathering team takes ~530 cycles per point --
// 10 VALU for addresses / weight broadcasts, 8 ds_read_b128, the LDS round trip, 16 v_pk_fma_f32, strictly one after the other --
// and with 1.5 gathering waves per SIMD nothing fills the round trips: LDS 24 % busy, VALU 41 %, the gathering half (17.1 K
// cycles) 1.5x the preparing one, a third of all wave time at the swap barrier.  The registers for a deeper pipeline were not
// there at three waves per SIMD (168 VGPRs).  Here
//   * a block is eight waves = two per SIMD = 256 VGPRs per wave, a team is FOUR waves (64 (query, head) slots: the 170 queries of
//     an item take three passes: level-0 rows 0-3 | rows 4-7 | the coarser levels), and wave w sits on SIMD w % 4: every SIMD
//     holds exactly one gathering and one preparing wave at any time -- one loads the LDS pipe, the other the VALU;
//   * the gather keeps TWO points in flight per wave: the eight reads of point j + 1 are on their way while the sixteen packed
//     multiply-adds of point j issue (two register sets, the LDS returns a wave's reads in order: s_waitcnt lgkmcnt(8) releases
//     the older set); the pipeline runs across the levels of a pass (the first two points of the next staged level are requested
//     in front of the last two multiply-add groups of the current one);
//   * a single wave issues at most one instruction every ~5 cycles (profiles/r03_valu_issue_probe.txt), two waves per SIMD
//     saturate the simple opcodes: the per-wave instruction count of a point (~35) is what paces the gathering half now.
// Everything else -- the arena shared from both ends, one late level per item, window DMA level by level with hardware zero fill,
// team-local meeting points, one block barrier per half period, exact-pyramid / nested-maps instantiations -- is generation 8's.
//
// Reference semantics: ms_deform_im2col_cuda.cuh:236-321 (forward), :30-86 (bilinear with zero padding).
#include "common.hpp"
#include <stdlib.h>
#include "kernels.hpp"
#include "msda_sample.hpp"
#include "msda_tiled6_helpers.hpp"

// Round 6: the build switches of rounds 4-5 are frozen at the values that won (instruction diet of the preparing half: all of
// it but the register-carried slot words; global-memory levels two points deep in flight; plain output stores) and the timing-only
// ablation builds are gone -- the losing sides and the ablation code are in the git history (`git log -- msda_tiled9.hip`,
// profiles/r05_msda_diet.txt has their numbers); tools/freeze_ifdefs.py is how the conditionals were resolved.
constexpr int T9_EARLY = 3;     // levels of pass 0 a team gathers at the END of its preparing half (behind a meeting point of its own): 410.9 vs 432.0 us with one
constexpr int T9_WAITCNT = 8;   // s_waitcnt lgkmcnt(N) that releases the older register set: 8 by construction
constexpr int T9_TW = 6;        // waves per team: 12 waves, 3 per SIMD, 2 passes of 96 slots
constexpr int T9_GPRIO = 2;     // s_setprio of a wave while it gathers

namespace vllm {

namespace {

__device__ unsigned long long g_t9_prof[16];
#define T9_TICK(slot)                                                            \
    if (PROF) {                                                                  \
        const unsigned now__ = (unsigned)__builtin_amdgcn_s_memtime();           \
        pacc[slot] += now__ - tprev;                                             \
        tprev = now__;                                                           \
    }

// the eight 16-byte pieces of one sampling point's 2 x 2 footprint that a lane owns (two channel chunks x four corners)
struct T9Set {
    float4_t a1, a2, a3, a4, c1, c2, c3, c4;
};
__device__ __forceinline__ void t9_read(T9Set &s, int b0, int b0p, int b1, int b1p)
{
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\t"
                 "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:128\n\t"
                 "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:128\n\t"
                 "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:128"
                 : "=&v"(s.a1), "=&v"(s.a2), "=&v"(s.a3), "=&v"(s.a4), "=&v"(s.c1), "=&v"(s.c2), "=&v"(s.c3), "=&v"(s.c4)
                 : "v"(b0), "v"(b0p), "v"(b1), "v"(b1p));
    // (no "memory" clobber: with it the compiler takes the statement for a possible reader of an LDS-DMA in flight and puts
    //  s_waitcnt vmcnt(0) in front of a pass's first read -- the late level's DMA and the previous pass's output stores would
    //  have to retire first.  The statements are volatile: they keep their order among themselves and against the waits /
    //  meeting points / barriers, which do carry the clobber; the windows they read were staged before such a point.)
}
// the set whose reads were issued BEFORE the newest eight has landed (the LDS returns a wave's reads in order) / everything has
template <int CNT>
__device__ __forceinline__ void t9_wait(T9Set &s)
{
    static_assert(CNT == 0 || CNT == T9_WAITCNT, "two register sets of eight reads each");
    if constexpr (CNT != 0)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(s.a1), "+v"(s.a2), "+v"(s.a3), "+v"(s.a4), "+v"(s.c1), "+v"(s.c2), "+v"(s.c3), "+v"(s.c4) : "n"(T9_WAITCNT));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s.a1), "+v"(s.a2), "+v"(s.a3), "+v"(s.a4), "+v"(s.c1), "+v"(s.c2), "+v"(s.c3), "+v"(s.c4));
}
// The sums of a (query, head) slot: four INDEPENDENT accumulation chains (two channel pairs of each of the lane's two 16-byte
// chunks), interleaved corner by corner -- a chain at a time (the first build's order) every packed multiply-add waits for the
// one in front of it: 16 dependent issues per point, ~340 cycles per point measured.  Per channel the corners are added in the
// order 1, 2, 3, 4, as in every other kernel of the family: same bits.
struct T9Acc {
    float2_t t0, t1, u0, u1;
};
__device__ __forceinline__ void t9_fma(T9Acc &a, const T9Set &s, float e1, float e2, float e3, float e4)
{
#define T9_CORNER(E, A, C)                                                                              \
    a.t0 = t6_fma2(E, (float2_t){A[0], A[1]}, a.t0); a.u0 = t6_fma2(E, (float2_t){C[0], C[1]}, a.u0);   \
    a.t1 = t6_fma2(E, (float2_t){A[2], A[3]}, a.t1); a.u1 = t6_fma2(E, (float2_t){C[2], C[3]}, a.u1);
    T9_CORNER(e1, s.a1, s.c1) T9_CORNER(e2, s.a2, s.c2) T9_CORNER(e3, s.a3, s.c3) T9_CORNER(e4, s.a4, s.c4)
#undef T9_CORNER
    asm volatile("" : "+v"(a.t0), "+v"(a.t1), "+v"(a.u0), "+v"(a.u1));
}

// EXACT: the level maps are exact halves (H_l << l == H_0): sizes by shifts; otherwise nested maps, sizes from LDS (generation 8).
template <int WIN, bool PROF, bool EXACT>
__global__ __launch_bounds__(T9_TW * 128, 1) void msda_fwd_tiled9_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ attw, int B, int S, int M, int L, int Lq,
    float *__restrict__ out, uint16_t *__restrict__ out16, int hinted)
{
    constexpr int D = 32, PT = 4, TW = T9_TW, NW = 2 * TW, NP = TW == 4 ? 3 : 2, THREADS = NW * 64, R = WIN;
    static_assert(TW == 4 || TW == 6, "teams of four or six waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_box = reinterpret_cast<int *>(smem + (T6_ZPX + R + T6_SLACK) * 128);   // [2 teams][4 levels][4]: min hl, min -hl, min wl, min -wl
    int *s_used = s_box + 32;                                                      // [2 teams]: pixels of the arena the team's item occupies
    int *s_cnt = s_box + 34;                                                       // [2 meeting points][2 teams]: arrivals (monotonic)

    if (EXACT ? !geometry_is_pyramid(shapes, L, Lq) : !(geometry_is_nested(shapes, L, Lq) && !geometry_is_pyramid(shapes, L, Lq))) {
        if (hinted) __builtin_trap();   // a stale "pyramid" hint must fail loudly, not leave `out` unwritten
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave_s >= TW ? 1 : 0;        // waves 0-3 / 4-7: one wave of each team per SIMD
    const int wt = wave_s - team * TW;            // wave within the team
    unsigned pacc[16] = {};
    unsigned tprev = PROF ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    const unsigned MD = (unsigned)(M * D);
    const int H0 = (int)shapes[0], W0 = (int)shapes[1];
    int *s_dim = s_box + 40;                      // nested maps: H[4], W[4], first query[4]
    if (!EXACT && tid < 4) {
        int q0 = 0;
        for (int l = 0; l < tid; ++l) q0 += (l < L) ? (int)shapes[2 * l] * (int)shapes[2 * l + 1] : 0;
        s_dim[tid] = tid < L ? (int)shapes[2 * tid] : 1;
        s_dim[4 + tid] = tid < L ? (int)shapes[2 * tid + 1] : 1;
        s_dim[8 + tid] = q0;
    }
    const int ntx0 = (W0 + 15) >> 4;
    const int n_tiles = ((H0 + 7) >> 3) * ntx0;
    const unsigned n_items = (unsigned)(B * M * n_tiles);
    const int n_slots = L == 1 ? 128 : L == 2 ? 160 : L == 3 ? 168 : 170;

    // ---- per-lane constants (as generations 6-8).  They are RE-DERIVED from the lane number inside every section that uses
    // them (a handful of VALU), through an opaque copy the optimiser cannot hoist: computed once in front of the item loop they
    // are ~50 registers that live through the gather, where two register sets of reads in flight need the room (the first
    // build spilled 113 registers: per-pass point data parked in scratch memory, reloads between the window DMA loops) ----
    auto lane_now = [&]() { int x = lane; asm volatile("" : "+v"(x)); return x; };
    auto chunk_of = [](int ln) { return ((((ln >> 2) >> 1) & 1) * 4 + (ln & 3)) * 16; };   // cA: byte offset of the lane's first channel chunk
    auto sinfo_of = [&](int p, int ln) {   // per pass, this lane's query slot: packed (level, y, x, dead)
        const int quad = ln >> 2;
        const int qslot = (quad & 8) | ((0x46751320 >> ((quad & 7) * 4)) & 7);
        const int s = p * (TW * 16) + wt * 16 + qslot;
        const int rr = min((s >= 128) + (s >= 160) + (s >= 168), 3);
        const int lo = s - (rr == 0 ? 0 : rr == 1 ? 128 : rr == 2 ? 160 : 168);
        return rr | ((lo >> (4 - rr)) << 2) | ((lo & ((16 >> rr) - 1)) << 6) | ((s >= n_slots ? 1 : 0) << 10);
    };
    const int v0k = (int)lsi[min(lane & 3, L - 1)];   // (the one per-lane constant that is a memory load: carried)

    for (int i = tid; i < T6_ZPX * 32; i += THREADS) reinterpret_cast<float *>(smem)[i] = 0.f;
    if (tid < 32) s_box[tid] = T6_BIG;
    if (tid < 2) s_used[tid] = 0;
    if (tid < 6) s_cnt[tid] = 0;
    __syncthreads();

    const unsigned xcd = blockIdx.x & 7;
    const unsigned ipx = (n_items + 7) >> 3;
    const unsigned blocks_per_xcd = gridDim.x >> 3;
    const unsigned j0 = blockIdx.x >> 3;
    // items of this block: xcd * ipx + j0 + i * blocks_per_xcd, i = 0 .. n_blk - 1; team t owns the i with i % 2 == t
    const unsigned lim = xcd * ipx < n_items ? min(ipx, n_items - xcd * ipx) : 0u;
    const int n_blk = j0 < lim ? (int)((lim - j0 + blocks_per_xcd - 1) / blocks_per_xcd) : 0;
    if (n_blk == 0) return;   // (block-uniform)

    auto pair_of = [&](int si, int b, int m, int ty, int tx, bool &ok) -> unsigned {
        const int sr = si & 3, sy = (si >> 2) & 15, sx = (si >> 6) & 15;
        const int y = ((ty * 8) >> sr) + sy, x = ((tx * 16) >> sr) + sx;
        const int HW0 = H0 * W0;
        const int sH = EXACT ? H0 >> sr : s_dim[sr], sW = EXACT ? W0 >> sr : s_dim[4 + sr];   // (once per item and pass)
        const int sQ = EXACT ? (sr >= 1 ? HW0 : 0) + (sr >= 2 ? HW0 >> 2 : 0) + (sr >= 3 ? HW0 >> 4 : 0) : s_dim[8 + sr];
        ok = !(si >> 10) && y < sH && x < sW;
        const int q = ok ? sQ + y * sW + x : (ty * 8) * W0 + tx * 16;
        return (unsigned)((b * Lq + q) * M + m);
    };
    // Round 5: the three divisions of an item number (block-uniform, once per item and wave: ~25 scalar / vector instructions each
    // through the compiler's reciprocal sequence) by multiply-high with magic numbers made once per launch.  floor(n / d) ==
    // mulhi(n, floor(2^32 / d) + 1) for n * d < 2^32; every dividend here is < n_items and n_items * max(d) < 2^32 is checked
    // (otherwise the magic numbers are 0 and the plain divisions run).
    const bool fastdiv = (unsigned long long)n_items * (unsigned)max(max(n_tiles, M), ntx0) < (1ull << 32);
    const unsigned mg_tiles = fastdiv ? (unsigned)(0xffffffffu / (unsigned)n_tiles) + 1u : 0u;
    const unsigned mg_M = fastdiv ? (unsigned)(0xffffffffu / (unsigned)M) + 1u : 0u;
    const unsigned mg_ntx = fastdiv ? (unsigned)(0xffffffffu / (unsigned)ntx0) + 1u : 0u;
    auto udiv = [](unsigned n, unsigned d, unsigned mg) { return mg ? __umulhi(n, mg) : n / d; };
    auto decode = [&](unsigned item, int &b, int &m, int &ty, int &tx) {
        const unsigned bm = udiv(item, (unsigned)n_tiles, mg_tiles), t = item - bm * (unsigned)n_tiles;
        const unsigned bb = udiv(bm, (unsigned)M, mg_M), yy = udiv(t, (unsigned)ntx0, mg_ntx);
        b = __builtin_amdgcn_readfirstlane((int)bb); m = __builtin_amdgcn_readfirstlane((int)(bm - bb * (unsigned)M));
        ty = __builtin_amdgcn_readfirstlane((int)yy); tx = __builtin_amdgcn_readfirstlane((int)(t - yy * (unsigned)ntx0));
    };
    auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };

    // ---- "next": the team's item whose locations / weights are in flight ----
    int i_next = team;
    bool nv = false;
    int nb = 0, nm = 0;
    unsigned npr[NP] = {};
    bool nqok[NP] = {};
    float4_t lc0[NP], lc1[NP], la[NP];
    auto prefetch_next = [&]() {
        nv = uni(i_next) < n_blk;
        if (nv) {
            int ty, tx;
            decode(xcd * ipx + j0 + (unsigned)uni(i_next) * blocks_per_xcd, nb, nm, ty, tx);
            const int ln = lane_now();
            const int kk = min(ln & 3, L - 1);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                npr[p] = pair_of(sinfo_of(p, ln), nb, nm, ty, tx, nqok[p]);
                const unsigned e = (npr[p] * (unsigned)L + (unsigned)kk) * PT;
                lc0[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2);
                lc1[p] = *reinterpret_cast<const float4_t *>(loc + (size_t)e * 2 + 4);
                la[p] = *reinterpret_cast<const float4_t *>(attw + (size_t)e);
            }
        }
        i_next += 2;
    };

    // ---- "cur": the team's item between P1 and the last gather pass ----
    bool cv = false;
    int cb = 0, cm = 0;
    unsigned prc[NP] = {};
    bool qokc[NP] = {};
    float w1[NP][4] = {}, w2[NP][4] = {}, w3[NP][4] = {}, w4[NP][4] = {};
    int o[NP][4] = {};
    [[maybe_unused]] unsigned okm[NP] = {};
    int4 bx = {0, 0, 0, 0};
    int lay = 0;
    float acc_e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // T9_EARLY: pass 0's sums of the levels gathered before the swap

    // Gather of one pass of the 16 (query, head) slots of this wave: the levels whose layout word says `want` (1: staged with the
    // item, 5: the item's late level), software-pipelined two points deep; want == 1 also takes the levels that come from global
    // memory.  A lane holds the point data of ITS level (k); lane LQ of a quad broadcasts them to the quad by DPP.
    auto gather = [&](const float (&w1c)[4], const float (&w2c)[4], const float (&w3c)[4], const float (&w4c)[4], const int (&oc)[4],
                      float (&acc)[8], int want, int lmin, int lmax) {
        const float *vbc = value + ((size_t)cb * S * M + cm) * D;
        const int ln = lane_now();
        const int cA = chunk_of(ln), cA0 = cA + (int)lds_addr(smem);
        // (timing-only ablation 64: every quad reads a fixed pixel whose parity is its quad number's -- the four quads of a
        //  ds_read_b128 lane group then hit four different bank sets: what a conflict-free window layout would buy)
        // (wave-uniform) which levels this call gathers, and their window pitch in bytes
        bool act[4];
        int pit[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            act[l] = ((__builtin_amdgcn_readlane(lay, l) >> 24) & 5) == want && l >= lmin && l < lmax;
            pit[l] = ((-__builtin_amdgcn_readlane(bx.w, l) + 1) - __builtin_amdgcn_readlane(bx.z, l) + 1) * 128;
        }
        T9Set sa, sb;
        T9Acc ac = {{acc[0], acc[1]}, {acc[2], acc[3]}, {acc[4], acc[5]}, {acc[6], acc[7]}};
        // One pipeline step: the multiply-adds of point (LQ, I_) from SET, then -- NX -- the reads of point (NLQ, NI) into the same
        // SET.  The next point's addresses and this point's weight broadcasts are evaluated IN FRONT of the wait (pinned by the
        // empty asm: a pure instruction is otherwise placed just above its first user), the wait releases SET when at most CNT
        // newer reads are outstanding.
#define T9_STEP(SET, I_, LQ, CNT, NX, NI, NLQ)                                                                    \
    {                                                                                                            \
        int b0_ = (NX) ? qbi<NLQ>(oc[NI]) + cA0 : 0, b1_ = b0_ ^ 64;                                               \
        int b0p_ = b0_ + pit[NLQ], b1p_ = b1_ + pit[NLQ];                                                        \
        float e1 = qbf<LQ>(w1c[I_]), e2 = qbf<LQ>(w2c[I_]), e3 = qbf<LQ>(w3c[I_]), e4 = qbf<LQ>(w4c[I_]);        \
        asm volatile("" : "+v"(b0_), "+v"(b1_), "+v"(b0p_), "+v"(b1p_), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4)); \
        t9_wait<CNT>(SET);                                                                                       \
        t9_fma(ac, SET, e1, e2, e3, e4);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (NX) t9_read(SET, b0_, b0p_, b1_, b1p_);                                                              \
    }
#define T9_RD(SET, I_, LQ)                                                                                       \
    {                                                                                                            \
        const int b0_ = qbi<LQ>(oc[I_]) + cA0, b1_ = b0_ ^ 64;                                                   \
        t9_read(SET, b0_, b0_ + pit[LQ], b1_, b1_ + pit[LQ]);                                                    \
    }
        // NO control flow between a read and the wait that releases it.  A first version requested the next level's first two
        // points under `if (next level is staged)` and carried them across the level blocks' joins: where the register allocator
        // had picked different registers for a set on the two sides of a join it inserted v_mov copies of registers whose reads
        // were STILL IN FLIGHT (to the compiler an asm statement's outputs are ready when it returns) -- stale sums for a wave's
        // whole pass, a few times in a thousand launches, only on inputs with late / global-memory levels (the paths with joins).
        // So: (1) all four levels staged (the common case): ONE straight-line pipeline over the 16 points of the pass; (2) anything
        // else: a self-contained pipeline per staged level (it drains at the level's end: one LDS round trip per level).
        {
            if (act[0] && act[1] && act[2] && act[3]) {
                T9_RD(sa, 0, 0) T9_RD(sb, 1, 0)
                T9_STEP(sa, 0, 0, T9_WAITCNT, true, 2, 0) T9_STEP(sb, 1, 0, T9_WAITCNT, true, 3, 0)
                T9_STEP(sa, 2, 0, T9_WAITCNT, true, 0, 1) T9_STEP(sb, 3, 0, T9_WAITCNT, true, 1, 1)
                T9_STEP(sa, 0, 1, T9_WAITCNT, true, 2, 1) T9_STEP(sb, 1, 1, T9_WAITCNT, true, 3, 1)
                T9_STEP(sa, 2, 1, T9_WAITCNT, true, 0, 2) T9_STEP(sb, 3, 1, T9_WAITCNT, true, 1, 2)
                T9_STEP(sa, 0, 2, T9_WAITCNT, true, 2, 2) T9_STEP(sb, 1, 2, T9_WAITCNT, true, 3, 2)
                T9_STEP(sa, 2, 2, T9_WAITCNT, true, 0, 3) T9_STEP(sb, 3, 2, T9_WAITCNT, true, 1, 3)
                T9_STEP(sa, 0, 3, T9_WAITCNT, true, 2, 3) T9_STEP(sb, 1, 3, T9_WAITCNT, true, 3, 3)
                T9_STEP(sa, 2, 3, T9_WAITCNT, false, 0, 3) T9_STEP(sb, 3, 3, 0, false, 0, 3)
            } else {
#define T9_LEVEL(LQ)                                                                                             \
    if (act[LQ]) {                                                                                               \
        T9_RD(sa, 0, LQ) T9_RD(sb, 1, LQ)                                                                        \
        T9_STEP(sa, 0, LQ, T9_WAITCNT, true, 2, LQ) T9_STEP(sb, 1, LQ, T9_WAITCNT, true, 3, LQ)                  \
        T9_STEP(sa, 2, LQ, T9_WAITCNT, false, 0, LQ) T9_STEP(sb, 3, LQ, 0, false, 0, LQ)                         \
    }
                T9_LEVEL(0) T9_LEVEL(1) T9_LEVEL(2) T9_LEVEL(3)
#undef T9_LEVEL
            }
        }
#undef T9_STEP
#undef T9_RD
        acc[0] = ac.t0.x; acc[1] = ac.t0.y; acc[2] = ac.t1.x; acc[3] = ac.t1.y; acc[4] = ac.u0.x; acc[5] = ac.u0.y; acc[6] = ac.u1.x; acc[7] = ac.u1.y;
        // cold levels: from global memory, the owner lane's point data by ds_bpermute (run-time level)
        for (int l = 0; l < (want == 1 && lmax == 4 ? L : 0); ++l) {
            const int lay_l = __builtin_amdgcn_readlane(lay, l);
            if (!((lay_l >> 25) & 1)) continue;
            const int Hc = EXACT ? H0 >> l : uni(s_dim[l]), Wc = EXACT ? W0 >> l : uni(s_dim[4 + l]);
            const float *vc = vbc + (size_t)__builtin_amdgcn_readlane(v0k, l) * MD;
            const int src = ((ln & ~3) | l) << 2;
            // Round 5: the four points of a cold level as a two-deep pipeline -- the 8 loads of point i + 1 are in flight under the
            // multiply-adds of point i (two register sets of 32: the hot gather's sets are dead here).  The rolled loop of round 4
            // paid one global-memory round trip per point, 4 per level and pass, and a cold item (6.8 % of the bench's items) took two to
            // three half periods.  Same expressions in the same order per point: same bits.
            struct ColdPt { float4_t a1, a2, a3, a4, d1, d2, d3, d4; float f1, f2, f3, f4; int m; };
            auto cold_issue = [&](ColdPt &c, int oci, float w1i, float w2i, float w3i, float w4i) {
                const int hwp = __builtin_amdgcn_ds_bpermute(src, oci);
                const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w1i)));
                const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w2i)));
                const float e3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w3i)));
                const float e4 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w4i)));
                const int bh = (hwp >> 16) - 1, bw = (hwp & 0xffff) - 1;
                const bool u0 = bh >= 0, u1 = bh + 1 <= Hc - 1, l0 = bw >= 0, l1 = bw + 1 <= Wc - 1;
                const int h0 = min(max(bh, 0), Hc - 1), h1 = min(max(bh + 1, 0), Hc - 1);
                const int c0 = min(max(bw, 0), Wc - 1), c1 = min(max(bw + 1, 0), Wc - 1);
                const float *p1 = vc + (size_t)((unsigned)(h0 * Wc + c0) * MD), *p2 = vc + (size_t)((unsigned)(h0 * Wc + c1) * MD);
                const float *p3 = vc + (size_t)((unsigned)(h1 * Wc + c0) * MD), *p4 = vc + (size_t)((unsigned)(h1 * Wc + c1) * MD);
                const int eA = cA / 4, eB = (cA ^ 64) / 4;
                c.a1 = load4(p1 + eA); c.a2 = load4(p2 + eA); c.a3 = load4(p3 + eA); c.a4 = load4(p4 + eA);
                c.d1 = load4(p1 + eB); c.d2 = load4(p2 + eB); c.d3 = load4(p3 + eB); c.d4 = load4(p4 + eB);
                c.m = ((u0 && l0) ? 1 : 0) | ((u0 && l1) ? 2 : 0) | ((u1 && l0) ? 4 : 0) | ((u1 && l1) ? 8 : 0);
                c.f1 = (u0 && l0) ? e1 : 0.f; c.f2 = (u0 && l1) ? e2 : 0.f; c.f3 = (u1 && l0) ? e3 : 0.f; c.f4 = (u1 && l1) ? e4 : 0.f;
            };
            auto cold_consume = [&](const ColdPt &c) {
                const bool k1 = c.m & 1, k2 = c.m & 2, k3 = c.m & 4, k4 = c.m & 8;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    acc[ch] += c.f1 * (k1 ? c.a1[ch] : 0.f) + c.f2 * (k2 ? c.a2[ch] : 0.f) + c.f3 * (k3 ? c.a3[ch] : 0.f) + c.f4 * (k4 ? c.a4[ch] : 0.f);
                    acc[4 + ch] += c.f1 * (k1 ? c.d1[ch] : 0.f) + c.f2 * (k2 ? c.d2[ch] : 0.f) + c.f3 * (k3 ? c.d3[ch] : 0.f) + c.f4 * (k4 ? c.d4[ch] : 0.f);
                }
            };
            {
                ColdPt ca, cb;
                cold_issue(ca, oc[0], w1c[0], w2c[0], w3c[0], w4c[0]);
                cold_issue(cb, oc[1], w1c[1], w2c[1], w3c[1], w4c[1]);
                __builtin_amdgcn_sched_barrier(0);
                cold_consume(ca);
                __builtin_amdgcn_sched_barrier(0);
                cold_issue(ca, oc[2], w1c[2], w2c[2], w3c[2], w4c[2]);
                __builtin_amdgcn_sched_barrier(0);
                cold_consume(cb);
                __builtin_amdgcn_sched_barrier(0);
                cold_issue(cb, oc[3], w1c[3], w2c[3], w3c[3], w4c[3]);
                __builtin_amdgcn_sched_barrier(0);
                cold_consume(ca);
                __builtin_amdgcn_sched_barrier(0);
                cold_consume(cb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PROF) pacc[12] += 1;
        }
    };
    // Output of one pass.  The stores are INLINE ASSEMBLY: for a compiler-visible store this compiler put s_waitcnt vmcnt(0) in
    // front of the next pass's first instruction that rewrites one of the store's address / data registers (seen in the ISA: the
    // next pass started behind the acknowledgement of these stores and behind the late level's DMA).  The hardware reads a
    // store's registers at issue; they are left alone for 16 wait states behind it (the margin gemm256p.hip measured).  The
    // compiler's own counted vmcnt waits stay correct: stores it does not know about can only make a wait longer.
    auto store_out = [&](const float (&acc)[8], bool qok, unsigned pr) {
        if (qok) {
            const int cA = chunk_of(lane_now());
            if (out16) {   // the caller (the fused layer) wants the bf16 operand of output_proj
                uint16_t *op = out16 + (size_t)pr * D;
                uint2_t o1, o2;
                o1.x = pack_bf16x2(acc[0], acc[1]); o1.y = pack_bf16x2(acc[2], acc[3]);
                o2.x = pack_bf16x2(acc[4], acc[5]); o2.y = pack_bf16x2(acc[6], acc[7]);
                uint16_t *p1 = op + cA / 4, *p2 = op + (cA ^ 64) / 4;
                *reinterpret_cast<uint2_t *>(p1) = o1;
                *reinterpret_cast<uint2_t *>(p2) = o2;
            } else {
                float *op = out + (size_t)pr * D;
                const float4_t v1 = {acc[0], acc[1], acc[2], acc[3]}, v2 = {acc[4], acc[5], acc[6], acc[7]};
                float *p1 = op + cA / 4, *p2 = op + (cA ^ 64) / 4;
                store4(p1, v1);
                store4(p2, v2);
            }
        }
    };
    // window DMA of ONE level: its np pixels (a multiple of 8) to arena pixel `base`, in 8-pixel groups (1 KiB per wave instruction);
    // this wave takes the groups g with (g + phase) % TW == its index in the team.  Everything per level -- box, pitch, magic,
    // buffer descriptor (out-of-map pixels get an offset beyond it: hardware zero fill) -- is set up once.
    auto dma_level = [&](int l, int np, int base, const float *vb, unsigned magick_, int phase) {
        const int y0 = __builtin_amdgcn_readlane(bx.x, l), x0 = __builtin_amdgcn_readlane(bx.z, l);
        const int ww = (-__builtin_amdgcn_readlane(bx.w, l) + 1) - x0 + 1;
        const unsigned magic = (unsigned)__builtin_amdgcn_readlane((int)magick_, l);
        const int Hl = EXACT ? uni(H0) >> l : uni(s_dim[l]), Wl = EXACT ? uni(W0) >> l : uni(s_dim[4 + l]);
        const uint64_t lvl = (uint64_t)(uintptr_t)vb + (uint64_t)(unsigned)__builtin_amdgcn_readlane(v0k, l) * (uint64_t)uni((int)MD) * 4u;
        const uint64_t lvl_u = ((uint64_t)(unsigned)uni((int)(lvl >> 32)) << 32) | (unsigned)uni((int)(unsigned)lvl);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)lvl_u, 0,
                                                                             (int)(((unsigned)(Hl * Wl - 1) * (unsigned)uni((int)MD) + 32u) * 4u), 0x00020000);
        int g0 = wt - phase;            // first group of this wave: (g0 + phase) % TW == wt
        g0 += g0 < 0 ? TW : 0;
        char *dst0 = smem + (size_t)(T6_ZPX + base) * 128;
        // (the lane number is re-derived here, two VALU: held in a register across the item loop it was spilled, and its reload
        //  between two levels' DMA loops came with an s_waitcnt vmcnt(0) -- a level's window had to land before the next one's
        //  requests went out)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int sub8 = ln & 7;
        // Round 5, second step: a wave takes a CONTIGUOUS run of 8-pixel groups of the concatenated windows (call sites), here groups
        // [phase, np) of this level: with groups dealt round-robin every wave set up every level (4 x ~68 instructions per item,
        // twice the cost of the rounds themselves); a contiguous sixth of the list touches 1-2 levels.  The walk is the incremental
        // one below with a step of 8 pixels.
        (void)g0;
        {
            const int pixi = uni(phase) * 8 + (ln >> 3);
            const int wy0 = (int)(__umul24((unsigned)pixi, magic) >> 20);          // (pixi < 2^11, magic <= 2^20 + 1)
            int gx = x0 + (pixi - __mul24(wy0, ww));
            unsigned voff = ((unsigned)(__mul24(y0 + wy0, Wl) + gx) * MD + (unsigned)sub8 * 4u) * 4u;
            const int dq = (int)((8u * magic) >> 20), dr = 8 - dq * ww;             // (wave-uniform; dr < ww)
            const unsigned stepN = (unsigned)(dq * Wl + dr) * (unsigned)uni((int)MD) * 4u;
            const unsigned stepC = stepN + (unsigned)(Wl - ww) * (unsigned)uni((int)MD) * 4u;
            const int xend = x0 + ww;
            // (the three level constants of the carry live in VECTOR registers through the loop, made opaque: left to itself the
            //  compiler re-materialises them from scalar registers every round -- a select cannot take two scalar operands)
            int wwv = ww;
            unsigned sNv = stepN, sCv = stepC;
            asm volatile("" : "+v"(wwv), "+v"(sNv), "+v"(sCv));
            for (int g = uni(phase); g < np; ++g) {
                const unsigned vo = (unsigned)gx < (unsigned)Wl ? voff : 0xfffffff0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst0 + g * 1024), 16, (int)vo, 0, 0, 0);
                gx += dr;
                const bool carry = gx >= xend;
                gx -= carry ? wwv : 0;
                voff += carry ? sCv : sNv;
            }
        }
    };

    // a team's meeting point: wait until `target` arrivals have been counted.  Bounded: the four waves of a team always take the
    // same path, so the count always comes -- but a mistake here must end as a failed launch, not as a hung device
    auto meet = [&](int *cnt, int target) {
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) __builtin_trap();
        }
    };
    int epoch = 0;                // items this team has prepared
    int epoch_late = 0;           // ... of which had a late level
    int late_l = -1, late_np = 0, late_base = 0;   // the current item's late level (or -1), its window size and arena position
    unsigned magick_c = 0;        // pix / ww magic of this lane's level (window DMA), kept for the late level
    // A wave alternates between the preparing half (P1, P2) and the gathering half of ITS team's items; team 1 runs half a period
    // behind team 0.  The two halves are STRAIGHT-LINE code in one loop body (prepare | barrier | gather | barrier), team 1 enters
    // through one extra barrier and team 0 leaves through one: as two arms of one branch under a loop header (generation 8's form)
    // the compiler has to assume any order of halves, which keeps the point data AND the locations in flight AND the gather's
    // register sets live everywhere -- at 256 registers that form spilled more than a hundred of them.
    prefetch_next();                       // the team's first item (team 0: item 0, team 1: item 1)
    if (team == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int n_iter = (n_blk + 1) >> 1;
#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
        T9_TICK(0)   // barrier + loop control
        {
            // ================= P1: cur <- next; this lane's 4 points of level k, all passes; boxes =================
            cv = nv; cb = nb; cm = nm;
#pragma unroll
            for (int p = 0; p < NP; ++p) { prc[p] = npr[p]; qokc[p] = nqok[p]; }
            // (the previous item's point data are dead here; they are only rewritten under `cv` below: without this the register
            //  allocator carries all 60 of them through the preparing half next to the 36 registers of the locations in flight)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(w1[p][i]), "=v"(w2[p][i]), "=v"(w3[p][i]), "=v"(w4[p][i]), "=v"(o[p][i]));
            const int lnA = lane_now();
            const int k = lnA & 3;                       // the value level this lane owns
            if (cv) {
                const int kk = min(k, L - 1);
                const int Hk = EXACT ? H0 >> kk : s_dim[kk], Wk = EXACT ? W0 >> kk : s_dim[4 + kk];
                // Round 5 form of the point arithmetic (profiles/r05_msda_diet.txt: 53 -> ~34 instructions per point).  Same values
                // as sample_point() for every accepted point: the acceptance test is the reference's four comparisons, floor() is
                // taken once and reused as the float the fractions are measured from ((float)(int)floor(x) == floor(x) for the
                // coordinates a map can have), the integer parts only ever reach the packed word `o`.  What changed:
                //   * ONE select per quantity that feeds the weights (the two fractions and the attention weight become 0 for a
                //     rejected point, all four products are then exactly 0) instead of one per weight and per integer part;
                //   * a rejected point is o = -1 (no separate mask word; the offsets below test the sign);
                //   * the box: both coordinates of `o` at once -- v_pk_min_u16 (a rejected point is 0xffff, 0xffff: never the minimum)
                //     and v_pk_max_i16 (a rejected point is -1, -1: never the maximum; accepted fields are 0 .. H), no selects, no
                //     negations; the packed pair goes through the two DPP steps and is unpacked by the 16 lanes that do the ds_min.
                int rmin = -1, rmax = -1;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const bool lane_ok = qokc[p] && k < L;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lx = i < 2 ? lc0[p][2 * i] : lc1[p][2 * i - 4], ly = i < 2 ? lc0[p][2 * i + 1] : lc1[p][2 * i - 3];
                        const float h_im = sub_rn(mul_rn(ly, (float)Hk), 0.5f), w_im = sub_rn(mul_rn(lx, (float)Wk), 0.5f);
                        const bool ok = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hk) && (w_im < (float)Wk) && lane_ok;
                        const float hf = floorf(h_im), wf = floorf(w_im);
                        const float lh = ok ? h_im - hf : 0.f, lw = ok ? w_im - wf : 0.f;
                        const float a = ok ? la[p][i] : 0.f;
                        const float hh = 1.f - lh, hw_ = 1.f - lw;
                        // (six products instead of eight: the attention weight goes into the two row factors first.  One rounding
                        //  placed differently from the other kernels of the family, 1 ulp of a weight: inside the 2e-6 the tests
                        //  allow between kernels, 4e-6 against the oracle)
                        const float ta = hh * a, tb = lh * a;
                        w1[p][i] = ta * hw_; w2[p][i] = ta * lw; w3[p][i] = tb * hw_; w4[p][i] = tb * lw;
                        const int oo = ok ? ((int)hf << 16) + (int)wf + 0x10001 : -1;
                        o[p][i] = oo;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(rmin) : "v"(rmin), "v"(oo));
                        asm("v_pk_max_i16 %0, %1, %2" : "=v"(rmax) : "v"(rmax), "v"(oo));
                    }
                }
                {
                    int t;
                    t = __builtin_amdgcn_update_dpp(rmin, rmin, 0x124, 0xf, 0xf, false); asm("v_pk_min_u16 %0, %1, %2" : "=v"(rmin) : "v"(rmin), "v"(t));
                    t = __builtin_amdgcn_update_dpp(rmin, rmin, 0x128, 0xf, 0xf, false); asm("v_pk_min_u16 %0, %1, %2" : "=v"(rmin) : "v"(rmin), "v"(t));
                    t = __builtin_amdgcn_update_dpp(rmax, rmax, 0x124, 0xf, 0xf, false); asm("v_pk_max_i16 %0, %1, %2" : "=v"(rmax) : "v"(rmax), "v"(t));
                    t = __builtin_amdgcn_update_dpp(rmax, rmax, 0x128, 0xf, 0xf, false); asm("v_pk_max_i16 %0, %1, %2" : "=v"(rmax) : "v"(rmax), "v"(t));
                }
                if ((lnA & 12) == 0 && rmax >= 0) {       // (rmax < 0: no accepted point of this level in this lane row)
                    const unsigned a = lds_addr(s_box + team * 16 + k * 4);
                    const int r0 = (int)((unsigned)rmin >> 16) - 1, r1 = 1 - (rmax >> 16);
                    const int r2 = (rmin & 0xffff) - 1, r3 = 1 - (rmax & 0xffff);
                    asm volatile("ds_min_i32 %0, %1\n\tds_min_i32 %0, %2 offset:4\n\tds_min_i32 %0, %3 offset:8\n\tds_min_i32 %0, %4 offset:12"
                                 :: "v"(a), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
                }
            }
            T9_TICK(1)
            // ================= P2: layout beside the other team's item, LDS offsets, window DMA =================
            if (cv) {
                // the team's own meeting point (the boxes of all four waves are in): an LDS counter, not the block barrier -- the
                // other team is in the middle of its gather and must not be held up.  The LDS executes a wave's operations in
                // order, so a wave that sees the full count also sees every minimum that was issued in front of an arrival.
                ++epoch;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(s_cnt + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                meet(s_cnt + team, epoch * TW);
                asm volatile("" ::: "memory");
                T9_TICK(8)   // team meeting point
                bx = *reinterpret_cast<const int4 *>(s_box + team * 16 + k * 4);   // lane l < 4: the box of level l
                const int used_other = uni(s_used[team ^ 1]);
                const bool anyk = bx.x != T6_BIG && k < L;
                const int wwk = (-bx.w + 1) - bx.z + 1;
                int np8k = anyk ? ((((-bx.y + 1) - bx.x + 1) * wwk + 7) & ~7) : 0;
                if (anyk && wwk > T6_ZPX - 2) np8k = 0x10000;
                // floor(2^20 / ww) + 1 through the hardware reciprocal: 2^20 / ww is an integer (ww a power of two: exact in float) or
                // at least 1 / ww away from one on either side, the float result is within 2^20 / ww * 2^-22 = 1 / (4 ww) of it --
                // the same number as the integer division (~20 instructions with four quarter-rate multiplies) for every ww >= 1
                const unsigned magick = (unsigned)(1048576.f * __builtin_amdgcn_rcpf((float)max(wwk, 1))) + 1u;
                int cum[5] = {0, 0, 0, 0, 0};
                magick_c = magick;
                {
                    // Levels are placed on the team's side of the arena beside what the other team's item occupies NOW.  ONE level
                    // that does not fit there becomes LATE: it is placed behind the team's other levels in the space the other
                    // team's item leaves at the swap, reserved here (s_used) so that the other team's next layout keeps clear of
                    // it, staged at the start of the gathering half and gathered behind a second meeting point.
                    const int limit = R - used_other;
                    int used = 0, lays[4];
                    late_l = -1; late_np = 0;
                    // Round 5: the common case -- every level fits beside the other team's item -- without the level-by-level
                    // scalar walk (~120 scalar instructions per wave and item): the inclusive prefix sum of the four window sizes
                    // on lanes 0 .. 3 (two DPP row shifts), one comparison of the total, bases / layout words computed on those
                    // lanes and read back with eight v_readlane.  Anything else (a level that does not fit: late or from global
                    // memory; a window wider than the zero strip, np8k = 0x10000 > limit) takes the walk below.
                    int incl = np8k;
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1, lanes without a source add 0
                    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                    const int total = __builtin_amdgcn_readlane(incl, 3);
                    const bool all_fit = total <= limit;
                    if (all_fit) {
                        const int basev = team ? R - incl : incl - np8k;
                        const int layv = np8k > 0 ? (basev | (1 << 24)) : 0;
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            lays[l] = __builtin_amdgcn_readlane(layv, l);
                            cum[l + 1] = __builtin_amdgcn_readlane(incl, l);
                        }
                        used = total;
                    } else
                    {
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int np = __builtin_amdgcn_readlane(np8k, l);
                        const bool fits = np > 0 && used + np <= limit;
                        const bool late = !fits && np > 0 && np <= R && late_l < 0;
                        const int base = team ? R - used - np : used;
                        lays[l] = fits ? (base | (1 << 24)) : (np > 0 ? (1 << 25) : 0);
                        used += fits ? np : 0;
                        cum[l + 1] = used;
                        late_np = late ? np : late_np;
                        late_l = late ? l : late_l;
                    }
                    if (late_l >= 0 && used + late_np <= R) {
                        late_base = team ? R - used - late_np : used;
                        const int v = late_base | (5 << 24);
                        if (late_l == 0) lays[0] = v; else if (late_l == 1) lays[1] = v; else if (late_l == 2) lays[2] = v; else lays[3] = v;
                        used += late_np;
                    } else {
                        late_l = -1;
                    }
                    }
                    lay = sel4(k, lays[0], lays[1], lays[2], lays[3]);
                    if (tid == team * (TW * 64)) s_used[team] = used;
                }
                {
                    const int y0k = bx.x, x0k = bx.z;
                    const int basek = lay & 0xffff;
                    const bool hotk = (lay >> 24) & 1;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            // (row * pitch by the 24-bit multiply-add -- full rate, the 32-bit product is quarter rate; the
                            //  constants of the level folded into one; rejected points carry a negative word)
                            const int cst = T6_ZPX + basek - (y0k + 1) * wwk - (x0k + 1);
                            const int lin = __mul24((int)((unsigned)o[p][i] >> 16), wwk) + (o[p][i] & 0xffff);   // (hl + 1) * ww + (wl + 1)
                            const int off = o[p][i] >= 0 ? (lin + cst) * 128 : 0;
                            o[p][i] = hotk ? off : o[p][i];
                        }
                }
                T9_TICK(2)   // layout + offsets
                // window DMA: the hot windows are ONE concatenated list of 8-pixel groups, group g belongs to wave g % TW of the team
                {
                    const float *vbn = value + ((size_t)cb * S * M + cm) * D;
                    const int n8 = uni(cum[4]) >> 3;                                   // groups of 8 pixels over all staged levels
                    const int ga = (wt * n8) / TW, gb = ((wt + 1) * n8) / TW;           // this wave's run
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const int c0 = uni(cum[l]) >> 3, c1 = uni(cum[l + 1]) >> 3;
                        const int lo = max(ga, c0), hi = min(gb, c1);
                        if (lo < hi) dma_level(l, hi - c0, __builtin_amdgcn_readlane(lay, l) & 0xffff, vbn, magick, lo - c0);
                    }
                }
                T9_TICK(3)   // DMA issue
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the windows has landed
                T9_TICK(4)   // DMA wait
                if (T9_EARLY > 0) {
                    // The preparing half is the shorter one: the team starts on ITS OWN gather as soon as its windows are in -- a second
                    // meeting point of the team (every wave's share of the windows has landed), no block barrier -- with the first
                    // T9_EARLY levels of pass 0 (staged ones; a late level and anything from global memory wait for the swap).  The
                    // eight partial sums travel across the swap barrier into pass 0.
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(s_cnt + 4 + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    meet(s_cnt + 4 + team, epoch * TW);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc_e[c] = 0.f;
                    gather(w1[0], w2[0], w3[0], w4[0], o[0], acc_e, 1, 0, T9_EARLY);
                    T9_TICK(11)   // early gather
                }
            }
        }
        // (a compiler-VISIBLE wait: the waits above are inline assembly, and on the path "locations requested, P1 skipped" -- which
        //  does not exist, but the compiler cannot know -- their loads would still be pending where the gather reuses the registers:
        //  it put s_waitcnt vmcnt(0) in front of every pass's first read, i.e. behind the late level's DMA)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        T9_TICK(0)
        {
            // ================= the three passes of the gather =================
            if (tid - team * (TW * 64) < 16) s_box[team * 16 + (tid - team * (TW * 64))] = T6_BIG;   // read in P2, written again in the next P1
            T9_TICK(5)
            if (cv) {
                if (T9_GPRIO) __builtin_amdgcn_s_setprio(T9_GPRIO);   // the gathering wave goes first on the SIMD it shares with a preparing one
                const bool has_late = uni(late_l) >= 0;   // (team-uniform)
                if (has_late) {          // its DMA goes out first and lands under pass 0 of the other levels
                    const float *vbc = value + ((size_t)cb * S * M + cm) * D;
                    const int n8l = uni(late_np) >> 3;
                    dma_level(uni(late_l), ((wt + 1) * n8l) / TW, uni(late_base), vbc, magick_c, (wt * n8l) / TW);
                }
                // The passes in a ROLLED loop over ONE copy of the gather (two with the late form): the gather always reads the
                // registers of pass 0, and behind a pass the next one's point data move there (20 moves; pass 0's are dead by then.
                // Indexing the per-pass arrays with a run-time pass number would put them in scratch memory, three unrolled copies
                // of the gather would be ~60 KB of code, copies into a fourth register set cost the allocator its budget).
#pragma unroll 1
                for (int p = 0; p < NP; ++p) {
                    float acc[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = (T9_EARLY > 0 && p == 0) ? acc_e[c] : 0.f;
                    // (the late form is a second trip through the same code)
#pragma unroll 1
                    for (int lt = 0; lt < (has_late ? 2 : 1); ++lt) {
                        if (lt == 1 && p == 0) {
                            T9_TICK(6)
                            ++epoch_late;
                            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the late window has landed
                            if (lane == 0) __hip_atomic_fetch_add(s_cnt + 2 + team, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            meet(s_cnt + 2 + team, epoch_late * TW);
                            asm volatile("" ::: "memory");
                            if (PROF) pacc[13] += 1;
                            T9_TICK(9)   // late level: meeting point
                        }
                        gather(w1[0], w2[0], w3[0], w4[0], o[0], acc, lt ? 5 : 1, (T9_EARLY > 0 && p == 0 && lt == 0) ? T9_EARLY : 0, 4);
                    }
                    store_out(acc, qokc[0], prc[0]);
#define T9_TAKE(P_) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { w1[0][i] = w1[P_][i]; w2[0][i] = w2[P_][i]; w3[0][i] = w3[P_][i]; w4[0][i] = w4[P_][i]; o[0][i] = o[P_][i]; } qokc[0] = qokc[P_]; prc[0] = prc[P_]; }
                    if (p == 0) T9_TAKE(1) else if (NP > 2 && p == 1) T9_TAKE(NP > 2 ? 2 : 1)
#undef T9_TAKE
                }
                T9_TICK(6)
                if (T9_GPRIO) __builtin_amdgcn_s_setprio(0);
            }
            T9_TICK(7)
            // (the locations / weights consumed in P1 are dead through the gather, but prefetch_next rewrites them only under `nv`:
            //  without this definition the allocator carries the 36 registers through the gather)
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("" : "=v"(lc0[p]), "=v"(lc1[p]), "=v"(la[p]));
            prefetch_next();   // the team's next item: its locations / weights travel across the barrier into P1
            if (PROF && cv) pacc[14] += 1;
            T9_TICK(10)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (team == 0) __syncthreads();        // (pairs with team 1's entry barrier)
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_t9_prof[i], (unsigned long long)pacc[i]);
    }
}

template <int WIN, bool PROF, bool EXACT>
int t9_go(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw, int B, int S,
          int M, int L, int Lq, float *out, uint16_t *out16, int hinted, hipStream_t st)
{
    const int cus = device_cus();   // (per device: a process-wide cache of the first device's count sized the grid wrongly on mixed hosts)
    constexpr size_t lds = (size_t)(T6_ZPX + WIN + T6_SLACK) * 128 + 256;
    static_assert(lds <= 163840, "LDS budget");
    static unsigned long long attr_mask = 0;
    if (first_use_on_device(&attr_mask)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_tiled9_kernel<WIN, PROF, EXACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    VLLM_LAUNCH((msda_fwd_tiled9_kernel<WIN, PROF, EXACT>), dim3((cus / 8) * 8), dim3(T9_TW * 128), lds, st, value, shapes, lsi, loc, attw,
                B, S, M, L, Lq, out, out16, hinted);
    VLLM_CHECK_LAUNCH("msda_fwd_tiled9_kernel");
    return VLLM_OK;
}

}  // namespace

// which: 1 the exact-pyramid instantiation, 2 the nested-maps one (each returns at once on maps that are not its own), 3 both
int msda_tiled9_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, int prof, hipStream_t st, uint16_t *out16, int hinted, int which)
{
    if (which & 1) {
        const int e = prof ? t9_go<1200, true, true>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st)
                           : t9_go<1200, false, true>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
        if (e) return e;
    }
    if (which & 2) {
        const int e = prof ? t9_go<1200, true, false>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st)
                           : t9_go<1200, false, false>(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, out16, hinted, st);
        if (e) return e;
    }
    return VLLM_OK;
}

int msda9_debug_counters(long *out, int n)
{
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t9_prof), sizeof(h)) != hipSuccess) {
        set_error("msda9_debug_counters: device read failed");
        return VLLM_ELAUNCH;
    }
    for (int i = 0; i < n && i < 16; ++i) out[i] = (long)h[i];
    const unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_t9_prof), z, sizeof(z));
    return n < 16 ? n : 16;
}


}  // namespace vllm
