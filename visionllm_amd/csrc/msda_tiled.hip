// MSDA forward: dispatcher of the LDS-tiled kernels for the encoder self-attention case (queries == pyramid pixels, Lq == S).
//
// Which kernel serves which call (round 5: ONE forward kernel per geometry class in the library; the other generations are in
// tools/experiments/ with their measurements in NOTES/):
//   * nested level maps (exact 2x pyramids and ceil- / floor-divided halves), fp32 values:  generation 9, msda_tiled9.hip;
//   * any other geometry, fp32 values:                                                      generation 4, msda_tiled4.hip;
//   * bf16 values (vllm_msda_forward_bf16, nested maps):                                    generation 6, msda_tiled6.hip;
//   * every other shape (decoder queries, D != 32, P != 4, fp64, tensors beyond 32-bit offsets): the gather kernels, msda.hip.
// The host may know the geometry (VLLM_GEO_*: one launch); if it does not, generation 9 and generation 4 are enqueued back to back
// and each checks the shape tensor on the device -- exactly one of them does the work, no host synchronisation.
#include "common.hpp"
#include "kernels.hpp"
#include "msda_sample.hpp"

namespace vllm {

constexpr int MT_MAXL = 8;

bool msda_tiled_ok(int D, int L, int P, int Lq, int S, int B, int M, const void *value, const void *out, const void *loc)
{
    // (the tiled kernels keep pixel / pair offsets in 32 bits: larger tensors take the gather kernel, whose indices are 64-bit)
    const bool fits32 = (long)S * M * 32 < (1L << 30) && (long)B * Lq * M * L * P * 2 < (1L << 30);
    return msda_tiled_enabled() && fits32 && D == 32 && P == 4 && L <= MT_MAXL && Lq == S && Lq >= 4096 && aligned16(value) && aligned16(out) &&
           (reinterpret_cast<uintptr_t>(loc) & 7u) == 0;
}

int msda_tiled4_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                       const float *attw, int B, int S, int M, int L, int Lq, float *out, int skip_pyramid,
                       hipStream_t st);   // msda_tiled4.hip
bool msda_tiled6_ok(int D, int L, int P, int Lq, int S, int B, int M);   // msda_tiled6.hip
int msda_tiled9_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, int prof, hipStream_t st, uint16_t *out16, int hinted,
                       int which);   // msda_tiled9.hip
int msda_tiled6_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, const float *attw,
                       int B, int S, int M, int L, int Lq, float *out, hipStream_t st);

// "msda_tiled" (vllm_set_option / VLLM_MSDA_TILED): 0 gather kernel, 1 automatic (default), 9 generation 4 forced (any geometry: the
// general-geometry kernel on every map), 5 generation 4 + phase clock, 2 / 8 generation 4's other configurations,
// 10-14 / 17 generation 6 on fp32 values (10 / 14 phase clock; its production use is the bf16-value operator),
// 20 generation 9 (= automatic on nested maps), 21 generation 9 + phase clock.
// Round 5: 3 (generation 2) and 18 / 19 (generation 8) left the library (tools/experiments/msda_tiled2.hip, msda_tiled8.hip); asking
// for them is an error, not a silent substitution.
// Automatic (1): generation 9 (pyramid items, two teams of six waves half a period apart) does the work when the
// level maps are nested halves -- it checks that on the device, from the shape tensor, and returns at once otherwise -- and the
// generation-4 launch behind it skips such maps, so exactly one of the two runs whatever the geometry, without a host
// synchronisation.
int msda_tiled_launch(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *attw, int B, int S, int M, int L, int Lq, int P, float *out, hipStream_t st, uint16_t *out16,
                      int *wrote16, int geometry)
{
    // out16 (optional): where a caller that wants the result in bf16 would like it.  *wrote16 = 1 tells it that a pyramid
    // geometry's result went THERE (and `out` was left alone); any other geometry's result is in `out` as usual.
    if (wrote16) *wrote16 = 0;
    const int mode = msda_tiled_enabled();
    VLLM_REQUIRE(mode != 3 && mode != 18 && mode != 19, "msda_tiled %d: that generation is not in the library any more (tools/experiments/)", mode);
    const bool fits32 = true;   // (msda_tiled_ok)
    // geometry (VLLM_GEO_*): what the HOST knows about the level maps.  UNKNOWN: both kernels are enqueued and the device
    // picks (no host synchronisation; one ~5 us empty launch); PYRAMID / GENERAL: exactly one launch.
    if ((mode == 1 || mode == 20 || mode == 21) && fits32 && msda_tiled6_ok(32, L, P, Lq, S, B, M) && aligned16(loc) && aligned16(attw) &&
        geometry != VLLM_GEO_GENERAL) {
        if (out16 && wrote16) *wrote16 = 1; else out16 = nullptr;
        // host hint: PYRAMID = exact halves, NESTED = halves rounded either way (one launch each); UNKNOWN: the device decides
        const int hinted = geometry == VLLM_GEO_PYRAMID || geometry == VLLM_GEO_NESTED;
        {
            const int which = geometry == VLLM_GEO_PYRAMID ? 1 : geometry == VLLM_GEO_NESTED ? 2 : 3;
            const int e = msda_tiled9_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, mode == 21, st, out16, hinted, which);
            if (e) return e;
        }
        if (hinted) return VLLM_OK;
        return msda_tiled4_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, 2, st);
    }
    if (mode >= 10 && fits32 && msda_tiled6_ok(32, L, P, Lq, S, B, M) && aligned16(loc) && aligned16(attw)) {
        if (int e = msda_tiled6_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, st)) return e;
        return msda_tiled4_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, 1, st);
    }
    return msda_tiled4_launch(value, shapes, lsi, loc, attw, B, S, M, L, Lq, out, 0, st);
}

}  // namespace vllm
