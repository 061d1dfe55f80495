"""Per-phase shader-clock breakdown of the LDS-tiled MSDA kernel (diagnostics build, option msda_tiled=5)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
names = ["item epilogue/stores", "A3: prefetch issue", "barrier B", "box read + exchange", "DMA issue", "offsets",
         "wait vmcnt(0)", "barrier C", "gather", "A1: point arithmetic", "A2: box reduction", "A0: level constants (LDS)", "item: decode of the next item", "count: levels gathered from global", "count: empty levels", "cold path (global gather / empty level)"]
_lib.set_option("msda_tiled", 5)
L = _lib.lib()
buf = (ctypes.c_long * 16)()
for rep in range(2):
    A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
    L.vllm_debug_counters(buf, 16)
tot = sum(buf[:13]) + buf[15]
for n, v in zip(names, buf[:16]):
    print("%-24s %12d  %5.1f%%" % (n, v, 100.0 * v / tot))
print("total ticks (sum over 512 blocks)", tot)
