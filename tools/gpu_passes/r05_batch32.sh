#!/bin/bash
# round 5, pass 32: point sampling with the channels-last LDS layout: tests + timing at the bench shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 600 python -m pytest tests/test_region_gpu.py -x -q 2>&1 | tail -3
timeout 300 python - <<'P' 2>&1 | grep -v amdgpu | tee gpurun_out/r05o/point_sample_cl.txt
import torch
from visionllm_amd import region_encoder as RE
dev = "cuda:0"; torch.manual_seed(0)
N, C, H, W, P = 16, 3072, 24, 24, 2304
x = torch.randn(N, C, H, W, device=dev); pts = torch.rand(N, P, 2, device=dev); valid = torch.rand(N, P, device=dev) < 0.5
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
us = t(lambda: RE.point_sample(x, pts)); ab = (x.numel() + pts.numel() + N * C * P) * 4
print(f"point_sample       {us:7.1f} us  {ab / us / 8e6:.3f} of 8 TB/s")
us = t(lambda: RE.point_sample_masked_mean(x, pts, valid)); ab = (x.numel() + pts.numel() + N * C) * 4
print(f"point_sample_mean  {us:7.1f} us  {ab / us / 8e6:.3f} of 8 TB/s")
P
