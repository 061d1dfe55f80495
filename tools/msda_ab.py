#!/usr/bin/env python3
"""Order-robust A/B of the MSDA forward kernel variants at cfg 4 (interleaved repeats, minimum per variant)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
def timeit(iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = {"gen4_win360_3blocks": 1, "gen4_win560_2blocks": 8, "gen4_8waves": 2, "gen5_4prod": 6, "gen2": 3, "gather": 0}
for _ in range(2):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); timeit(3)
best = {k: 1e9 for k in modes}
for _ in range(4):
    for k, v in modes.items():
        _lib.set_option("msda_tiled", v); best[k] = min(best[k], timeit())
print(json.dumps({k: round(v, 1) for k, v in best.items()}))
