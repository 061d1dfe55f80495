"""MSDA generation 6 (msda_tiled6.hip): per-phase shader-clock breakdown (option msda_tiled=10) and an interleaved A/B of
the gather-arithmetic variants (10 phase clock, 11 DPP moves + v_pk_fma_f32, 12 DPP moves + v_fma_f32) against the default
(v_fmac_f32_dpp) and generation 4."""
import ctypes, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
names7 = ["loop control + stores", "S1 wait prefetch + point arithmetic", "S2 box reduction", "wait own DMA", "barrier X",
          "S3/S5 layout + offsets", "decode + prefetch issue", "S6 gather + interleaved DMA", "cold levels", "remaining DMA issue",
          "-", "-", "count: cold levels", "-", "count: items", "-"]
names = ["stores + loop control", "S1 wait prefetch + point arithmetic", "S2 box reduction", "barrier B", "S3/S5 layout + offsets",
         "S4 DMA issue", "wait own DMA", "barrier C", "next item decode + prefetch issue", "S6 gather", "cold levels",
         "-", "count: cold levels", "count: groups", "count: items", "-"]
L = _lib.lib()
def run():
    return A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
PM = int(os.environ.get("T6_PROF_MODE", "10"))
_lib.set_option("msda_tiled", PM)
buf = (ctypes.c_long * 16)()
for rep in range(2):
    run()
    L.vllm_debug_counters(buf, 16)
tot = sum(buf[:11])
for n, v in zip(names7 if PM >= 15 else names, buf[:16]):
    print("%-40s %14d  %5.1f%%" % (n, v, 100.0 * v / tot if not n.startswith("count") else 0.0))
print("ticks per item (per reporting wave):", tot / max(buf[14], 1))
def timeit(iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = {"gen7_default": 1, "gen6": 17, "gen4_360_3": 9}
for _ in range(2):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); timeit(3)
best = {k: 1e9 for k in modes}
for _ in range(4):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); best[k] = min(best[k], timeit())
print(json.dumps({k: round(v, 1) for k, v in best.items()}))
