import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
libs = {}
for m in (0, 31):
    L = ctypes.CDLL(os.path.join(ROOT, "visionllm_amd", "_build_abl", f"libmsdabwd_abl{m}.so"))
    L.bt_abl_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
    libs[m] = L
dev = "cuda:0"
for shapes in (CFG4_SHAPES, [(72, 64), (36, 32), (18, 16), (9, 8)]):
  for B in (1, 2, 8):
    for mode in ("encoder_like", "uniform_like"):
        g = make_inputs(1, 8, 32, shapes, 4, mode="encoder_like", seed=0)
        t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
        for k in ("value", "loc", "attw"):
            t[k] = t[k].repeat(B, *([1] * (t[k].dim() - 1))).contiguous()
        if mode == "uniform_like":
            t["loc"] = torch.rand_like(t["loc"])
        _, S, M, D = t["value"].shape
        Lq, Lv = t["loc"].shape[1], t["loc"].shape[3]
        go = torch.randn(B, Lq, M * D, device=dev)
        gv, gl, gw = torch.zeros_like(t["value"]), torch.zeros_like(t["loc"]), torch.zeros_like(t["attw"])
        st = torch.cuda.current_stream().cuda_stream
        for m, L in libs.items():
            f = lambda: L.bt_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                                     go.data_ptr(), B, S, M, Lv, Lq, gv.data_ptr(), gl.data_ptr(), gw.data_ptr(), st)
            f(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize()
            print(f"S={S} B={B} {mode:13s} mask {m:2d}: {e0.elapsed_time(e1) / 3:8.3f} ms")
