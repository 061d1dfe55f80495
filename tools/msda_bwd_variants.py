"""Times builds of msda_bwd_mfma.hip (standalone -DBT_ABL_ENTRY libraries in visionllm_amd/_build_abl/libbwdv_*.so) at BASELINE cfg 4,
B = 8, encoder shape, interleaved on one box, and checks every build against the first one (grad_value 2^-18 of the magnitude sum,
per-point gradients 1e-4)."""
import ctypes, glob, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
names = sys.argv[1:] or sorted(os.path.basename(f)[len("libbwdv_"):-3] for f in glob.glob(os.path.join(ROOT, "visionllm_amd", "_build_abl", "libbwdv_*.so")))
libs = {}
for n in names:
    L = ctypes.CDLL(os.path.join(ROOT, "visionllm_amd", "_build_abl", f"libbwdv_{n}.so"))
    L.bt_abl_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
    libs[n] = L
dev = "cuda:0"
g = make_inputs(1, 8, 32, CFG4_SHAPES, 4, mode="encoder_like", seed=0)
t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
B = 8
for k in ("value", "loc", "attw"):
    t[k] = t[k].repeat(B, *([1] * (t[k].dim() - 1))).contiguous()
_, S, M, D = t["value"].shape
Lq, Lv = t["loc"].shape[1], t["loc"].shape[3]
go = torch.randn(B, Lq, M * D, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(L, gv, gl, gw):
    return L.bt_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                        go.data_ptr(), B, S, M, Lv, Lq, gv.data_ptr(), gl.data_ptr(), gw.data_ptr(), st)
ref = None
for n, L in libs.items():
    gv, gl, gw = torch.zeros_like(t["value"]), torch.zeros_like(t["loc"]), torch.zeros_like(t["attw"])
    rc = run(L, gv, gl, gw); torch.cuda.synchronize()
    if ref is None:
        ref = (gv, gl, gw); print(f"{n}: reference build rc {rc}  |gv| {gv.abs().sum().item():.6e}")
    else:
        ev = (gv - ref[0]).abs().max().item() / ref[0].abs().max().item()
        el = (gl - ref[1]).abs().max().item() / ref[1].abs().max().item()
        ew = (gw - ref[2]).abs().max().item() / ref[2].abs().max().item()
        print(f"{n}: rc {rc}  max diff / max: grad_value {ev:.2e} grad_loc {el:.2e} grad_attw {ew:.2e}  {'OK' if max(ev, el, ew) < 1e-4 else 'MISMATCH'}")
gv, gl, gw = torch.zeros_like(t["value"]), torch.zeros_like(t["loc"]), torch.zeros_like(t["attw"])
best = {n: 1e9 for n in libs}
for _ in range(4):
    for n, L in libs.items():
        run(L, gv, gl, gw)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run(L, gv, gl, gw)
        e1.record(); torch.cuda.synchronize()
        best[n] = min(best[n], e0.elapsed_time(e1) / 3)
for n in best:
    print(f"{n:24s} {best[n]:8.3f} ms")
