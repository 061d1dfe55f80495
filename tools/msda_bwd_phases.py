"""Phase clock of msda_bwd_tiled.hip (diagnostics build -DBT_PROF, built by the caller: see tools/gpu_passes) at BASELINE cfg 4, B = 8."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
name = sys.argv[1] if len(sys.argv) > 1 else "libmsdabwd_prof.so"
L = ctypes.CDLL(os.path.join(ROOT, "visionllm_amd", "_build_abl", name))
L.bt_abl_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
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
f = lambda: L.bt_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                         go.data_ptr(), B, S, M, Lv, Lq, gv.data_ptr(), gl.data_ptr(), gw.data_ptr(), st)
buf = (ctypes.c_long * 16)()
f(); torch.cuda.synchronize(); L.bt_abl_prof(buf)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1):.3f} ms")
L.bt_abl_prof(buf)
if "mfma" in name:
    names = ["item set-up (+ grad_out operand)", "level loop top", "A points", "window barrier", "C corner reads / grad_loc / grad_attw (+ wave minima)",
             "-", "D offset table + scatter + barrier", "D MFMA + flush atomics", "-", "-", "-", "D barrier between rounds"]
else:
    names = ["item set-up", "level hand-over (2 barriers)", "A points + wave reduction", "window barrier + clear + barrier", "C corner reads / gradients / LDS adds",
             "barrier behind C", "D flush (global atomics)"]
tot = sum(buf[i] for i in range(len(names)))
for i, n in enumerate(names):
    if n != "-": print(f"{n:40s} {buf[i]:14d} {100.0 * buf[i] / tot:6.1f} %")
c0, c1 = (9, 10) if "mfma" in name else (8, 9)
if "mfma" in name: tot -= buf[9] + buf[10]
print(f"level passes {buf[c0]}  mean window {buf[c1] / max(buf[c0], 1):.0f} px  ticks per level pass {tot / max(buf[c0], 1):.0f}")
