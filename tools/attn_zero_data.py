"""Is the attention kernel bound by the package power limit?  Same launches on random data, on zeros (no toggling in the
matrix pipe / operand buses: DVFS gives the clock back) and on constant ones."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
def t(qkv, n, S, H, D, var, reps=30):
    out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    _lib.set_option("attn_variant", var)
    best = 1e9
    for _ in range(5):
        for _ in range(3): _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(reps): _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for (n, S, H, D) in ((40, 577, 16, 64), (40, 1025, 25, 128)):
    for name, q in (("randn", torch.randn(n, S, 3, H, D, device="cuda")), ("zeros", torch.zeros(n, S, 3, H, D, device="cuda")),
                    ("small", torch.randn(n, S, 3, H, D, device="cuda") * 1e-3)):
        q = q.bfloat16()
        print(n, S, H, D, name, {v: round(t(q, n, S, H, D, v), 1) for v in (2, 74, 66)})
