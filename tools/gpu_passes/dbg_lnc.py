import math, sys, torch
sys.path.insert(0, ".")
from visionllm_amd import _lib
L = _lib.lib(); P = _lib.ptr; st = _lib.current_stream()
bf = lambda t: t.to(torch.bfloat16).contiguous()
torch.manual_seed(1)
M, C, N, eps = 23080, 1024, 3072, 1e-5
h = bf(torch.randn(M, C, device="cuda") * 2 + 0.75)
hb = h.float().view(M, 4, 256); mean = hb.mean(2)
stats = torch.stack([mean, ((hb - mean[..., None]) ** 2).sum(2)], 2).contiguous()
gamma = bf(1 + 0.2 * torch.randn(C, device="cuda")); w = bf(torch.randn(N, C, device="cuda") / 32)
wf = bf(w.float() * gamma.float()[None]); colsum = wf.float().sum(1).contiguous(); bias = torch.randn(N, device="cuda").contiguous()
ys = []
for fl in (0, 0x1000, 0):
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(L.vllm_gemm_bf16_ln(P(h), P(wf), None, P(y), M, N, C, C, C, N, 2 | fl, None, None, 0, None, P(stats), 4, 0, eps, P(colsum), P(bias), st))
    ys.append(y)
torch.cuda.synchronize()
for i, y in enumerate(ys):
    nan = torch.isnan(y.float())
    print("leg", i, "nan count", int(nan.sum()), "persistent launches", L.vllm_gemm_persistent_launches())
    if nan.any():
        idx = nan.nonzero()
        print(" first", idx[:5].tolist(), "rows with nan", idx[:, 0].unique().numel(), "cols with nan", idx[:, 1].unique().numel())
        r = idx[:, 0]; c = idx[:, 1]
        print(" row%256 hist", torch.bincount(r % 256, minlength=256).nonzero().flatten()[:40].tolist())
        print(" col%256 hist", torch.bincount(c % 256, minlength=256).nonzero().flatten()[:40].tolist())
        print(" tiles", torch.unique(torch.stack([r // 256, c // 256], 1), dim=0)[:20].tolist())
d = (ys[0].float() - ys[1].float()).abs()
print("max diff persistent vs tilewise", d[~torch.isnan(d)].max().item(), "equal", torch.equal(ys[0], ys[1]), "run-to-run", torch.equal(ys[0], ys[2]))
bad = (ys[0] != ys[1])
if bad.any():
    idx = bad.nonzero(); r = idx[:, 0]; c = idx[:, 1]
    print("mismatch count", int(bad.sum()), "tiles", torch.unique(torch.stack([r // 256, c // 256], 1), dim=0)[:30].tolist())
    print(" row%256", torch.bincount(r % 256, minlength=256).nonzero().flatten()[:64].tolist())
