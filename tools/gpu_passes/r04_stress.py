"""Round-4 race screen: the kernels added this round, many launches each on fresh outputs, every result compared with the first
(half-height GEMM tail, streaming K = 256 GEMMs incl. the layer's softmax / location epilogue, windowed DCNv3 backward)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from visionllm_amd import _lib, dcnv3 as DC, ms_deform_attn as A
L = _lib.lib(); st = _lib.current_stream(); P = _lib.ptr
dev = "cuda:0"
torch.manual_seed(0)
bad = 0
for name, M, N, K, epi in (("qkv40", 23080, 3072, 1024, 0), ("qkv32", 18464, 3072, 1024, 1), ("ivit_fc1_8t", 8200, 12800, 3200, 1), ("two_k_tiles", 2 * 256 * 4 + 130, 2048, 128, 0)):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    first = None
    n0 = L.vllm_gemm_half_tail_launches()
    for i in range(150):
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, None, 0, 0, st))
        if first is None: first = y
        elif not torch.equal(first, y): bad += 1; print("MISMATCH", name, i, float((first.float() - y.float()).abs().max()))
    print(name, "150 launches, half-tail launches", L.vllm_gemm_half_tail_launches() - n0, "finite", bool(torch.isfinite(first.float()).all()), flush=True)
for name, M, epi in (("skinny_bias", 70001, 0), ("skinny_f32", 70001, 5), ("skinny_f32_mask", 70000, 5)):
    K = N = 256
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.06).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    mask = (torch.rand(M, device=dev) < 0.2).to(torch.uint8) if "mask" in name else None
    first = None
    for i in range(150):
        y = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if epi == 5 else torch.bfloat16)
        _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, P(mask) if mask is not None else None, 0, 0, st))
        if first is None: first = y
        elif not torch.equal(first, y): bad += 1; print("MISMATCH", name, i)
    print(name, "150 launches ok", flush=True)
from msda_inputs import CFG4_SHAPES
B, C, Mh, Lv, Pp = 2, 256, 8, 4, 4
S = sum(h * w for h, w in CFG4_SHAPES)
mod = A.MSDeformAttn(C, Lv, Mh, Pp).to(dev).to(torch.bfloat16).eval()
ss = torch.tensor(CFG4_SHAPES, device=dev); lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
src = torch.randn(B, S, C, device=dev).bfloat16(); q = torch.randn(B, S, C, device=dev).bfloat16(); ref = torch.rand(B, S, Lv, 2, device=dev)
mask = torch.zeros(B, S, dtype=torch.bool, device=dev); mask[-1, -S // 7:] = True
first = None
with torch.no_grad():
    for i in range(60):
        o = mod(q, ref, src, ss, lsi, mask)
        if first is None: first = o.clone()
        elif not torch.equal(first, o): bad += 1; print("MISMATCH layer", i)
print("layer (streaming GEMMs + operator) 60 launches ok", flush=True)
N_, H, W, G, Cg, k = 2, 84, 84, 8, 32, 3
x = torch.randn(N_, H, W, G * Cg, device=dev); off = torch.randn(N_, H, W, G * k * k * 2, device=dev) * 2
m = torch.softmax(torch.randn(N_, H, W, G, k * k, device=dev), -1).reshape(N_, H, W, -1); go = torch.randn(N_, H, W, G * Cg, device=dev)
first = None
for i in range(60):
    gi, gof, gm = DC.dcnv3_backward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, Cg, 1.0, go)
    if first is None: first = (gi.clone(), gof.clone(), gm.clone())
    else:
        if not (torch.equal(first[1], gof) and torch.equal(first[2], gm)): bad += 1; print("MISMATCH dcnv3 offset/mask", i)
        if float((first[0] - gi).abs().max()) > 1e-4 * float(first[0].abs().max()): bad += 1; print("MISMATCH dcnv3 grad_input", i)
print("dcnv3 windowed backward 60 launches ok", flush=True)
print("BAD", bad)
