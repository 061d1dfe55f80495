"""LDS-tiled DCNv3 forward (dcnv3_tiled.hip): per-phase shader-clock breakdown (option dcnv3_tiled = 2)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionllm_amd import _lib, dcnv3 as A
names_old = ["loop control + store drain", "point arithmetic (+ wait for prefetched offsets)", "box reduction + barrier 1",
             "table + prefetch issue + DMA issue", "DMA wait + barrier 2", "gather + store issue", "count: tiles (x waves)", "-"]
names_pipe = ["-", "point arithmetic + box reduction", "wait (DMA, loads, stores) + barrier", "window geometry + offsets + prefetch issue",
              "gather + interleaved DMA issue", "remaining DMA issue + stores", "rotation + decode", "-", "count: tiles (x waves)", "-"]
MODE = int(os.environ.get("DCN_PROF_MODE", "2"))
names = names_pipe if MODE == 2 else names_old
N, H, W, G, C, k = 8, 168, 168, 20, 32, 3
torch.manual_seed(0)
x = torch.randn(N, H, W, G * C, device="cuda")
off = torch.randn(N, H, W, G * k * k * 2, device="cuda") * float(os.environ.get("DCN_OFFSET_SIGMA", "1.0"))
m = torch.softmax(torch.randn(N, H, W, G, k * k, device="cuda"), -1).reshape(N, H, W, -1)
_lib.set_option("dcnv3_tiled", MODE)
buf = (ctypes.c_long * 16)()
for rep in range(2):
    A.dcnv3_forward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0)
    _lib.lib().vllm_debug_counters(buf, 16)
cnt = buf[8] if MODE == 2 else buf[6]
tot = sum(buf[:7]) if MODE == 2 else sum(buf[:6])
for n, v in zip(names, buf[:10]):
    print("%-52s %14d  %5.1f%%" % (n, v, 100.0 * v / tot if not n.startswith("count") else 0.0))
print("ticks per tile (per reporting wave):", tot / max(cnt, 1))
