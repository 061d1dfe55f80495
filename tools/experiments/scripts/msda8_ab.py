#!/usr/bin/env python3
"""A/B of the MSDA forward generations 7 / 8 at cfg 4, batch 8 (interleaved repeats, minimum per variant) + generation 8's phase clock."""
import json, os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
def run():
    return A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
def timeit(iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = {"gen8": 18, "gen9": 20}   # (generation 7, mode 15, left the library in round 4: tools/experiments/msda_tiled7.hip)
_lib.set_option("msda_tiled", 0); ref = run()
for k, v in modes.items():
    _lib.set_option("msda_tiled", v); o = run(); torch.cuda.synchronize()
    print(k, "max abs diff vs gather kernel", float((o - ref).abs().max()), flush=True)
    timeit(3)
best = {k: 1e9 for k in modes}
for _ in range(5):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); best[k] = min(best[k], timeit())
print(json.dumps({k: round(v, 1) for k, v in best.items()}))
# phase clock of generation 8 (ticks of s_memtime summed over all waves)
_lib.set_option("msda_tiled", 19)
lib = _lib.lib()
buf = (ctypes.c_long * 16)()
lib.vllm_debug_counters(buf, 16)
run(); torch.cuda.synchronize()
n = lib.vllm_debug_counters(buf, 16)
names = ["barrier+loop", "P1 points+boxes", "P2 layout+offsets", "P2 dma issue", "P2 dma wait", "G head", "G0 gather", "G1 gather+prefetch",
         "P team meeting point", "G pass 0 late level + store", "-", "-", "cold levels", "late levels", "items", "-"]
tot = sum(buf[i] for i in range(12))
for i in range(16):
    if names[i] != "-": print(f"{names[i]:28s} {buf[i]:14d} {100.0 * buf[i] / max(tot, 1):6.1f} %" if i < 12 else f"{names[i]:28s} {buf[i]:14d}")
