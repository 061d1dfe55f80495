#!/bin/bash
# round 5, pass 28: the visual-token splice as one native call (slot scan on the device): tests + timing against the round-4 form
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 600 python -m pytest tests/test_vit_gpu.py tests/test_tokens_gpu.py -x -q -k "splice or token" 2>&1 | tail -15
timeout 300 python - <<'P' 2>&1 | grep -v amdgpu | tee gpurun_out/r05o/splice_native.txt
import torch, time
from visionllm_amd import splice as SP, _lib
dev = "cuda:0"
B, Lt, Cc, T = 8, 4096, 4096, 576
emb = torch.randn(B, Lt, Cc, device=dev).to(torch.bfloat16)
ids = torch.zeros(B, Lt, dtype=torch.int64, device=dev); ids[:, 100:100 + 5 * T] = 7
feats = torch.randn(B * 5, T, Cc, device=dev).to(torch.bfloat16)
def old():
    selected = ids == 7
    has_image = selected.sum(-1) != 0
    has_image = torch.cat([has_image[i][None].repeat(5) for i in range(B)], dim=0)
    vit = feats.reshape(-1, Cc) if bool(has_image.all()) else feats[has_image].reshape(-1, Cc)
    idx = torch.nonzero(selected.reshape(-1), as_tuple=False).reshape(-1)
    _lib.check(_lib.lib().vllm_scatter_rows_bf16(_lib.ptr(vit), _lib.ptr(idx), _lib.ptr(emb), idx.numel(), Cc, B * Lt, _lib.current_stream(emb.device)))
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ab = 2.0 * feats.numel() * 2
for name, f in (("round-4 form (torch mask / nonzero + scatter kernel)", old), ("native, checked", lambda: SP.splice_visual_tokens(emb, ids, 7, feats, [5] * B)),
                ("native, check=False", lambda: SP.splice_visual_tokens(emb, ids, 7, feats, [5] * B, check=False))):
    us = t(f); print(f"{name:55s} {us:8.1f} us  {ab / us / 1e3:7.1f} GB/s = {ab / us / 8e6:.3f} of 8 TB/s")
P
