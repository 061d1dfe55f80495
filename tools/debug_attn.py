import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from visionllm_amd import _lib
dev = "cuda:0"
L = _lib.lib(); st = _lib.current_stream(torch.device(dev))
def run(B, S, H, D, qkv):
    out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev)
    _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), B, S, H, D, D ** -0.5, st))
    return out
for D in (128,):
    S = 64
    # multiplicity of each key: batch b has v[key=b][:] = 64 -> out = multiplicity(b)
    qkv = torch.zeros(S, S, 3, 1, D, device=dev)
    for b in range(S):
        qkv[b, b, 2, 0, :] = 64.0
    out = run(S, S, 1, D, qkv.to(torch.bfloat16))
    mult = out[:, 0, 0, :].float()        # [key, d]
    print("D", D, "key multiplicities (d=0):", mult[:, 0].tolist())
    bad = (mult - 1).abs() > 0.01
    print("  bad (key,d) count", bad.sum().item(), "keys with any bad:", bad.any(1).nonzero().flatten().tolist())
    for k in bad.any(1).nonzero().flatten().tolist()[:6]:
        print("   key", k, "mult per d:", mult[k].tolist())
    # query-row dependence: same for other q rows?
    print("  q row 17, key mult d=0:", out[:, 17, 0, 0].float().tolist())
