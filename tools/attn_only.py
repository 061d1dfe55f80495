"""ViT-L attention shape (40 tiles x 16 heads x S577 x d64) run a few times -- a target for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
n, S, H, D = (8, 1025, 25, 128) if os.environ.get("ATTN_D") == "128" else (40, 577, 16, 64)
qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
for _ in range(10):
    _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
torch.cuda.synchronize()
