#!/usr/bin/env python3
"""DCNv3 forward micro-benchmark at InternImage-H-like stage shapes (channels 320 * 2^i, group_channels 16... here 32)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionllm_amd import dcnv3 as A  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    shapes = ((8, 336, 336, 10, 32), (8, 168, 168, 20, 32), (8, 84, 84, 40, 32), (8, 42, 42, 80, 32))
    if os.environ.get("DCN_C16"):   # group channels 16 (InternImage-T/S/B): 64 / 128 / 256 / 512 channels
        shapes = ((8, 336, 336, 4, 16), (8, 168, 168, 8, 16), (8, 84, 84, 16, 16), (8, 42, 42, 32, 16))
    for N, H, W, G, C in shapes:
        k = 3
        x = torch.randn(N, H, W, G * C, device=dev)
        off = torch.randn(N, H, W, G * k * k * 2, device=dev) * float(os.environ.get("DCN_OFFSET_SIGMA", "1.0"))
        m = torch.softmax(torch.randn(N, H, W, G, k * k, device=dev), -1).reshape(N, H, W, -1)
        sec = timeit(lambda: A.dcnv3_forward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0))
        algo = (x.numel() * 2 + off.numel() + m.numel()) * 4   # input + output + offsets + mask, fp32
        print(json.dumps(dict(N=N, H=H, W=W, G=G, C=C, us=sec * 1e6, algo_GBs=algo / sec / 1e9, frac_of_8TBs=algo / sec / 8e12)))
        if os.environ.get("DCN_BWD"):   # backward (round 4): input + grad_output read, grad_input accumulated, offsets / mask + their gradients
            go = torch.randn(N, H, W, G * C, device=dev)
            secb = timeit(lambda: A.dcnv3_backward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0, go), iters=5)
            algob = (x.numel() * 3 + 2 * (off.numel() + m.numel())) * 4
            print(json.dumps(dict(backward=True, N=N, H=H, W=W, G=G, C=C, us=secb * 1e6, algo_GBs=algob / secb / 1e9, frac_of_8TBs=algob / secb / 8e12)))


if __name__ == "__main__":
    main()
