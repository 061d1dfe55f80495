"""Same-box A/B of attention across builds of libvllm_hip.so: python tools/attn_ab_libs.py lib1.so lib2.so ...  Each build runs in its
own process (VLLM_HIP_LIB), the builds alternate for `ROUNDS` rounds, the minimum per build and shape is printed."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, os.environ["ROOT_"])
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
res = {}
for (n, S, H, D) in ((40, 577, 16, 64), (40, 1025, 25, 128)):
    qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
    out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    b = 1e9
    for _ in range(6):
        for _ in range(3): _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
        e1.record(); torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1) / 20 * 1e3)
    res[f"d{D}"] = round(b, 1)
    res[f"sum{D}"] = float(out.float().sum())
print(json.dumps(res))
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
            print(lib, "FAILED", o.stderr[-500:]); continue
        b = best.setdefault(lib, dict(d))
        for k, v in d.items():
            if k.startswith("d"): b[k] = min(b[k], v)
for lib in libs:
    print(lib, json.dumps(best.get(lib)))
