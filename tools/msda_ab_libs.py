"""Same-box A/B of the MSDA forward (cfg 4, B = 8, encoder shape, the bench's inputs) across builds of libvllm_hip.so:
    python tools/msda_ab_libs.py lib1.so lib2.so ...
Each build runs in its own process (VLLM_HIP_LIB), the builds alternate for ROUNDS rounds; minimum per build, and a digest of the
output (the builds must agree to fp32 rounding)."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
ROOT = os.environ["ROOT_"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
A.remember_geometry(t["shapes"])
f = lambda: A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
out = f(); torch.cuda.synchronize()
b = 1e9
for _ in range(6):
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    b = min(b, e0.elapsed_time(e1) / 10 * 1e3)
again = f()
print(json.dumps({"us": round(b, 1), "sum": float(out.double().sum()), "abs": float(out.double().abs().sum()), "stable": bool(torch.equal(out, again))}))
'''
libs = sys.argv[1:]
ROUNDS = 3
best = {}
for r in range(ROUNDS):
    for lib in libs:
        env = dict(os.environ, VLLM_HIP_LIB=os.path.join(ROOT, lib), ROOT_=ROOT)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        try:
            d = json.loads(o.stdout.strip().splitlines()[-1])
        except Exception:
            print(lib, "FAILED", o.stderr[-800:]); continue
        b = best.setdefault(lib, dict(d))
        b["us"] = min(b["us"], d["us"])
        b["stable"] = b["stable"] and d["stable"]
for lib in libs:
    print(lib, json.dumps(best.get(lib)))
