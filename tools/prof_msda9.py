"""MSDA generation 9 (msda_tiled9.hip): per-phase shader-clock breakdown (option msda_tiled = 21: the PROF instantiation, ticks of
every wave summed) at BASELINE cfg 4, B = 8, and the timing of the library's automatic choice / generation 8 / the gather kernel."""
import ctypes, json, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
names = ["barrier + loop control", "P1 point arithmetic + boxes", "P2 layout + offsets", "P2 DMA issue", "P2 DMA wait", "G clear boxes",
         "G gather passes + stores", "G after passes", "P team meeting point", "G late level meeting point", "G prefetch next item",
         "P early gather (incl. its meeting point)", "count: cold levels", "count: late levels", "count: items", "-"]
L = _lib.lib()
def run():
    return A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
_lib.set_option("msda_tiled", 21)
buf = (ctypes.c_long * 16)()
run(); L.vllm_debug_counters(buf, 16)
for rep in range(4):
    run()
L.vllm_debug_counters(buf, 16)
tot = sum(buf[:12])
for n, v in zip(names, buf[:16]):
    print("%-44s %14d  %5.1f%%" % (n, v, 100.0 * v / tot if not n.startswith("count") and n != "-" else 0.0))
print("ticks per wave and item:", tot / max(buf[14], 1))
def timeit(iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = {"automatic_gen9": 1, "gen4_any_geometry": 9, "gather": 0}
for _ in range(2):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); timeit(3)
best = {k: 1e9 for k in modes}
for _ in range(4):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); best[k] = min(best[k], timeit())
print(json.dumps({k: round(v, 1) for k, v in best.items()}))
