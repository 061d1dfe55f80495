import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
def t(M,N,K,epi=0,force=0):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, epi | force, None, None, 0, 0, st))
    best = 1e9
    for _ in range(5):
        for _ in range(5): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best
print(os.environ.get("VLLM_HIP_LIB", "in-tree build")[-34:], "4096^3 %.1f us   qkv %.1f  fc1 %.1f  fc2(192) %.1f us" % (
    t(4096, 4096, 4096), t(23080, 3072, 1024, 0), t(23080, 4096, 1024, 2), t(23080, 1024, 4096, 0)))
