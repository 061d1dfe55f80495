"""Times the ablation builds of msda_bwd_tiled.hip at BASELINE cfg 4 (B = 8, encoder shape): where the 22 ms go."""
import ctypes, glob, os, re, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
NAMES = {0: "full", 1: "no flush atomics", 2: "no direct (cold-window) atomics", 3: "no global atomics at all", 4: "no LDS adds",
         7: "no atomics, no LDS adds", 8: "no value corner reads", 15: "no atomics / LDS adds / corner reads", 31: "skeleton only", 63: "skeleton, no window clear", 95: "skeleton, no flush loop", 127: "skeleton, no clear, no flush loop"}
libs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "visionllm_amd", "_build_abl", "libmsdabwd_abl*.so"))):
    m = int(re.search(r"abl(\d+)\.so", f).group(1))
    L = ctypes.CDLL(f)
    L.bt_abl_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
    libs[m] = L
dev = "cuda:0"
g = make_inputs(1, 8, 32, CFG4_SHAPES, 4, mode="encoder_like", seed=0)
t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
B = 8
for k in ("value", "loc", "attw"):
    t[k] = t[k].repeat(B, *([1] * (t[k].dim() - 1))).contiguous()
_, S, M, D = t["value"].shape
Lq, Lv = t["loc"].shape[1], t["loc"].shape[3]
go = torch.randn(B, Lq, M * D, device=dev)
gv, gl, gw = torch.zeros_like(t["value"]), torch.zeros_like(t["loc"]), torch.zeros_like(t["attw"])
st = torch.cuda.current_stream().cuda_stream
best = {m: 1e9 for m in libs}
for _ in range(3):
    for m, L in libs.items():
        f = lambda: L.bt_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                                 go.data_ptr(), B, S, M, Lv, Lq, gv.data_ptr(), gl.data_ptr(), gw.data_ptr(), st)
        f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
        best[m] = min(best[m], e0.elapsed_time(e1) / 3)
for m in sorted(best):
    print(f"mask {m:3d} {NAMES.get(m, '?'):42s} {best[m]:8.3f} ms   (full - this = {best[0] - best[m]:7.3f})")
