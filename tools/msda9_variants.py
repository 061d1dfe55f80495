"""Times the side builds of msda_tiled9.hip (tools/msda9_variants.sh) at BASELINE cfg 4 (B = 8, encoder shape), interleaved over
rounds, next to the library's generation 8 / 9; builds whose name does not start with "abl" are also checked against the gather kernel."""
import ctypes, glob, os, re, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
libs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "visionllm_amd", "_build_abl", "libmsda9_*.so"))):
    name = re.search(r"libmsda9_(\w+)\.so", f).group(1)
    L = ctypes.CDLL(f)
    L.t9_abl_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
    libs[name] = L
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
B, S, M, D = t["value"].shape
Lq, Lv = t["loc"].shape[1], t["loc"].shape[3]
st = torch.cuda.current_stream().cuda_stream
_lib.set_option("msda_tiled", 0)
ref = A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
out = torch.empty_like(ref)
fns = {}
for name, L in libs.items():
    fns[name] = (lambda L=L: L.t9_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                                          B, S, M, Lv, Lq, out.data_ptr(), st))
def lib_mode(m):
    def f():
        _lib.set_option("msda_tiled", m)
        A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
    return f
fns["lib_gen9"] = lib_mode(20)   # (generation 8 left the library in round 5: tools/experiments/msda_tiled8.hip)
for name, f in fns.items():
    if name.startswith("abl") or name.startswith("lib"): continue
    out.zero_(); f(); torch.cuda.synchronize()
    print(name, "max abs diff vs gather kernel", float((out - ref).abs().max()), flush=True)
best = {m: 1e9 for m in fns}
for _ in range(5):
    for m, f in fns.items():
        for _ in range(2): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best[m] = min(best[m], e0.elapsed_time(e1) / 10 * 1e3)
for m in best:
    print(f"{m:24s} {best[m]:7.1f} us")
