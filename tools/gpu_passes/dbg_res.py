import math, sys, torch
sys.path.insert(0, ".")
from visionllm_amd import _lib
L = _lib.lib(); P = _lib.ptr; st = _lib.current_stream()
bf = lambda t: t.to(torch.bfloat16).contiguous()
torch.manual_seed(1)
M, N, K = 23080, 1024, 1024
x = bf(torch.randn(M, K, device="cuda")); w = bf(torch.randn(N, K, device="cuda") / 32); b = bf(torch.randn(N, device="cuda"))
res = bf(torch.randn(M, N, device="cuda"))
ys = []
for fl in (0, 0, 0x1000, 0):
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, 3 | fl, None, P(res), N, 0, st))
    ys.append(y)
torch.cuda.synchronize()
ref = (res.float() + x.float() @ w.float().t() + b.float())
for name, a_, b_ in (("run0 vs run1", ys[0], ys[1]), ("run0 vs run3", ys[0], ys[3]), ("persistent vs tilewise", ys[0], ys[2])):
    bad = a_ != b_
    print(name, "mismatches", int(bad.sum()))
    if bad.any() and int(bad.sum()) < 10_000_000:
        idx = bad.nonzero(); r = idx[:, 0]; c = idx[:, 1]
        print("  row%256", torch.bincount(r % 256, minlength=256).nonzero().flatten().tolist()[:80])
        print("  col%256", torch.bincount(c % 256, minlength=256).nonzero().flatten().tolist()[:80])
        print("  tiles", torch.unique(torch.stack([r // 192, c // 256], 1), dim=0)[:24].tolist(), "first", idx[:6].tolist())
for i, y in enumerate(ys):
    e = (y.float() - ref).abs()
    print("leg", i, "max err vs fp32 ref", e.max().item(), "count > 0.1:", int((e > 0.1).sum()))
    if (e > 0.1).any():
        idx = (e > 0.1).nonzero(); r = idx[:, 0]; c = idx[:, 1]
        print("  row%192", torch.bincount(r % 192, minlength=192).nonzero().flatten().tolist()[:80])
        print("  col%256", torch.bincount(c % 256, minlength=256).nonzero().flatten().tolist()[:80])
