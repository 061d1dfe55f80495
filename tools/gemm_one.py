"""One GEMM shape run a few times (a rocprofv3 --pmc target).  argv: M N K [lib]; VLLM_GEMM_VARIANT selects our schedule,
'lib' runs torch.nn.functional.linear (hipBLASLt) instead."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
M, N, K = (int(v) for v in sys.argv[1:4])
lib = len(sys.argv) > 4 and sys.argv[4] == "lib"
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
b = torch.zeros(N, device="cuda").bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
L = _lib.lib(); st = _lib.current_stream()
for _ in range(6):
    if lib:
        torch.nn.functional.linear(x, w, b)
    else:
        _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, 0, None, None, 0, 0, st))
torch.cuda.synchronize()
