import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionllm_amd import _lib
dev = "cuda:0"; L = _lib.lib(); st = _lib.current_stream(torch.device(dev)); P = _lib.ptr
M = N = K = 128
x = torch.zeros(M, K, device=dev); x[torch.arange(M), torch.arange(K)] = 1.0
w = (torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) % 251 / 251.0).to(torch.bfloat16)
y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
_lib.check(L.vllm_gemm_bf16(P(x.to(torch.bfloat16)), P(w), None, P(y), M, N, K, K, K, N, 0, None, None, 0, 0, st))
ref = w.float().t()
err = (y.float() - ref).abs()
print("gemm identity: max err", err.max().item(), "n bad", (err > 1e-3).sum().item())
bad = (err > 1e-3).nonzero()[:10]
for m, n in bad.tolist():
    print("  y[%d][%d]=%g ref %g" % (m, n, y[m, n].item(), ref[m, n].item()))
