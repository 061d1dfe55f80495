"""Same-box A/B of two builds of libvllm_hip.so (VLLM_HIP_LIB names the build): attention, the encoder GEMMs, norms.
One process per build; run the builds alternately and compare the minima."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()


def best(fn, rounds=5, reps=20):
    b = 1e9
    for _ in range(rounds):
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1) / reps * 1e3)
    return round(b, 1)


res = {"lib": os.environ.get("VLLM_HIP_LIB", "default")}
for (n, S, H, D) in ((40, 577, 16, 64), (8, 1025, 25, 128)):
    qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
    out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    res[f"attn_d{D}"] = best(lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st)))
for name, M, N, K, epi in (("qkv", 23080, 3072, 1024, 0), ("proj", 23080, 1024, 1024, 0), ("fc1", 23080, 4096, 1024, 2),
                           ("fc2", 23080, 1024, 4096, 0), ("ivit_fc1", 8200, 12800, 3200, 1)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res[name] = best(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, epi, None, None, 0, 0, st)))
x = torch.randn(23080, 1024, device="cuda").bfloat16(); w = torch.ones(1024, device="cuda").bfloat16(); y = torch.empty_like(x)
res["layernorm"] = best(lambda: _lib.check(L.vllm_layernorm_bf16(_lib.ptr(x), 1024, _lib.ptr(w), _lib.ptr(w), _lib.ptr(y), 1024, 23080, 1024, 1e-5, st)))
print(json.dumps(res))
