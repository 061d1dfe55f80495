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
int gemm_variant_override()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("VLLM_GEMM_VARIANT");
        v = e ? atoi(e) : 0;
        if (v < 0 || v > 2) v = 0;
    }
    return v;
}
}  // namespace vllm

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
