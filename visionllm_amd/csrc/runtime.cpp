// Library-wide runtime helpers: ABI version, per-thread last error, device info.
#include "common.hpp"
#include <string.h>
#include <stdlib.h>
#include "kernels.hpp"

namespace vllm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static int g_gemm_variant = -1, g_msda_tiled = -1, g_attn_variant = -1, g_gemm_direct = -1, g_layer_fused = -1;
int g_dcnv3_tiled = -1;
int dcnv3_tiled_enabled()
{
    if (g_dcnv3_tiled < 0) {   // 1 (default): LDS-tiled DCNv3 forward where it applies; 0: the gather kernel
        const char *e = getenv("VLLM_DCNV3_TILED");
        g_dcnv3_tiled = e ? atoi(e) : 1;
        if (g_dcnv3_tiled < 0 || g_dcnv3_tiled > 4) g_dcnv3_tiled = 1;
    }
    return g_dcnv3_tiled;
}
int msda_layer_fused()
{
    if (g_layer_fused < 0) {   // 1 (default): query GEMM with the sampling epilogue + bf16 operator output; 0: round-1 composition
        const char *e = getenv("VLLM_MSDA_LAYER_FUSED");
        g_layer_fused = e ? (atoi(e) != 0) : 1;
    }
    return g_layer_fused;
}
int gemm_direct_store()
{
    if (g_gemm_direct < 0) {   // 0 through LDS, 1 direct, 2 automatic (default)
        const char *e = getenv("VLLM_GEMM_DIRECT_STORE");
        g_gemm_direct = e ? atoi(e) : 2;
        if (g_gemm_direct < 0 || g_gemm_direct > 2) g_gemm_direct = 2;
    }
    return g_gemm_direct;
}
int attn_variant()
{
    if (g_attn_variant < 0) {
        const char *e = getenv("VLLM_ATTN_VARIANT");
        g_attn_variant = e ? (atoi(e) & 0xffff) : 32;
    }
    return g_attn_variant;
}
// persistent GEMM tile order (gemm256p.hip): -1 automatic (default), 0 the dense XCD order, RB > 0 banded with RB row panels per band
int g_gemm_tile_rb = -2;
int gemm_tile_rb()
{
    if (g_gemm_tile_rb == -2) {
        const char *e = getenv("VLLM_GEMM_TILE_RB");
        g_gemm_tile_rb = e ? atoi(e) : -1;
        if (g_gemm_tile_rb < -1 || g_gemm_tile_rb > 32) g_gemm_tile_rb = -1;
    }
    return g_gemm_tile_rb;
}
int gemm_variant_override()
{
    if (g_gemm_variant < 0) {
        const char *e = getenv("VLLM_GEMM_VARIANT");
        g_gemm_variant = e ? atoi(e) : 0;
        if (g_gemm_variant < 0 || g_gemm_variant > 4 || g_gemm_variant == 3) g_gemm_variant = 0;
    }
    return g_gemm_variant;
}
int msda_tiled_enabled()
{
    if (g_msda_tiled < 0) {
        const char *e = getenv("VLLM_MSDA_TILED");
        g_msda_tiled = e ? atoi(e) : 1;
        if (g_msda_tiled < 0 || g_msda_tiled > 21 || g_msda_tiled == 4 || g_msda_tiled == 6 || g_msda_tiled == 7 || g_msda_tiled == 15 || g_msda_tiled == 16) g_msda_tiled = 1;
    }
    return g_msda_tiled;
}
}  // namespace vllm

extern "C" int vllm_set_option(const char *name, int value)
{
    if (!name) return VLLM_EINVAL;
    if (!strcmp(name, "msda_tiled")) {
        const int old = vllm::msda_tiled_enabled();
        if (value < 0 || value > 21 || value == 4 || value == 6 || value == 7 || value == 15 || value == 16) {
            vllm::set_error("msda_tiled must be one of 0, 1, 2, 3, 5, 8, 9, 10..14, 17 .. 21");
            return VLLM_EINVAL;
        }
        vllm::g_msda_tiled = value;
        return old;
    }
    if (!strcmp(name, "dcnv3_tiled")) { const int old = vllm::dcnv3_tiled_enabled(); vllm::g_dcnv3_tiled = (value < 0 || value > 4) ? 1 : value; return old; }
    if (!strcmp(name, "dcnv3_bwd_tiled")) return vllm::dcnv3_bwd_tiled_set(value);
    if (!strcmp(name, "gemm_half_tail")) return vllm::gemm_half_tail_set(value);
    if (!strcmp(name, "msda_layer_value_bf16")) return vllm::msda_layer_value_bf16_set(value);
    if (!strcmp(name, "gemm_skinny")) return vllm::gemm_skinny_set(value);
    if (!strcmp(name, "msda_layer_fused")) { const int old = vllm::msda_layer_fused(); vllm::g_layer_fused = value != 0; return old; }
    if (!strcmp(name, "gemm_direct_store")) { const int old = vllm::gemm_direct_store(); vllm::g_gemm_direct = (value < 0 || value > 2) ? 2 : value; return old; }
    if (!strcmp(name, "attn_variant")) { const int old = vllm::attn_variant(); vllm::g_attn_variant = value & 0xffff; return old; }
    if (!strcmp(name, "gemm_tile_rb")) { const int old = vllm::gemm_tile_rb(); vllm::g_gemm_tile_rb = (value < -1 || value > 32) ? -1 : value; return old; }
    if (!strcmp(name, "gemm_variant")) {
        const int old = vllm::gemm_variant_override();
        if (value < 0 || value > 4 || value == 3) { vllm::set_error("gemm_variant must be 0, 1, 2 or 4"); return VLLM_EINVAL; }
        vllm::g_gemm_variant = value;
        return old;
    }
    vllm::set_error("unknown option %s", name);
    return VLLM_EINVAL;
}
namespace vllm { int dcnv3_pipe_debug_counters(long *out, int n); int gemm256_debug_counters(long *out, int n); int dcnv3_debug_counters(long *out, int n); int msda_debug_counters(long *out, int n); int msda6_debug_counters(long *out, int n); int msda9_debug_counters(long *out, int n); }
extern "C" int vllm_debug_counters(long *out, int n)
{
    if (!out || n <= 0) { vllm::set_error("vllm_debug_counters: bad arguments"); return VLLM_EINVAL; }
    { static const int gp = [] { const char *e = getenv("VLLM_GEMM_PROF"); return e ? atoi(e) : 0; }(); if (gp) return vllm::gemm256_debug_counters(out, n); }
    if (vllm::dcnv3_tiled_enabled() == 2) return vllm::dcnv3_pipe_debug_counters(out, n);
    if (vllm::dcnv3_tiled_enabled() == 4) return vllm::dcnv3_debug_counters(out, n);
    const int mode = vllm::msda_tiled_enabled();
    return mode >= 20 ? vllm::msda9_debug_counters(out, n) : mode >= 10 ? vllm::msda6_debug_counters(out, n) : vllm::msda_debug_counters(out, n);
}
// ---- in-step kernel timing ----------------------------------------------------------------------------------------
#include <vector>
#include <mutex>
namespace vllm {
int g_prof_on = 0;
namespace {
std::mutex g_prof_mu;
std::vector<hipEvent_t> g_prof_pool;            // events, reused across read-outs
std::vector<int> g_prof_tags;                   // tag of mark i (event i)
}
void prof_mark_slow(int tag, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const size_t i = g_prof_tags.size();
    if (i >= (1u << 20)) return;                // bounded: a forgotten vllm_prof_enable(1) must not eat the host's memory
    if (i >= g_prof_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_prof_pool.push_back(e);
    }
    if (hipEventRecord(g_prof_pool[i], st) == hipSuccess) g_prof_tags.push_back(tag);
}
}  // namespace vllm
static const char *const kProfNames[vllm::PT_COUNT] = {"end", "embed", "norm", "gemm_qkv", "qk_norm", "attn", "gemm_proj", "gemm_fc1",
                                                        "gemm_fc2", "gemm_bridge", "bridge_other", "msda_encoder_shape", "msda_other",
                                                        "msda_layer"};
extern "C" int vllm_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(vllm::g_prof_mu);
    vllm::g_prof_tags.clear();
    vllm::g_prof_on = on != 0;
    return vllm::PT_COUNT;
}
extern "C" const char *vllm_prof_tag_name(int tag) { return tag >= 0 && tag < vllm::PT_COUNT ? kProfNames[tag] : ""; }
extern "C" int vllm_prof_read(double *us_sum, long *count, int n)
{
    if (!us_sum || !count || n < vllm::PT_COUNT) { vllm::set_error("vllm_prof_read: need %d slots", (int)vllm::PT_COUNT); return VLLM_EINVAL; }
    std::lock_guard<std::mutex> lk(vllm::g_prof_mu);
    for (int i = 0; i < n; ++i) { us_sum[i] = 0.0; count[i] = 0; }
    const size_t m = vllm::g_prof_tags.size();
    if (m && hipEventSynchronize(vllm::g_prof_pool[m - 1]) != hipSuccess) { vllm::set_error("vllm_prof_read: event wait failed"); return VLLM_ELAUNCH; }
    for (size_t i = 0; i + 1 < m; ++i) {
        const int tag = vllm::g_prof_tags[i];
        if (tag == vllm::PT_END) continue;      // the gap between two C calls belongs to the host, not to a kernel
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, vllm::g_prof_pool[i], vllm::g_prof_pool[i + 1]) != hipSuccess) continue;
        us_sum[tag] += (double)ms * 1e3;
        count[tag] += 1;
    }
    vllm::g_prof_tags.clear();
    return vllm::PT_COUNT;
}
extern "C" int vllm_abi_version(void) { return VLLM_ABI_VERSION; }
extern "C" const char *vllm_last_error(void) { return vllm::g_err; }
extern "C" int vllm_device_info(char *name, int cap)
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        vllm::set_error("no HIP device");
        return VLLM_ELAUNCH;
    }
    if (name && cap > 0) {
        strncpy(name, prop.gcnArchName, cap - 1);
        name[cap - 1] = 0;
    }
    return prop.multiProcessorCount;
}
