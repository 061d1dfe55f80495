#!/usr/bin/env python3
"""Deformable-attention LAYER at BASELINE cfg 4 (B=8, d_model 256, 8 heads, 4 levels, 4 points): the fused native call
(vllm_msda_layer_forward) against the composed path (torch bf16 linears / softmax around the native operator)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES  # noqa: E402
from visionllm_amd import ms_deform_attn as A  # noqa: E402


def timeit(fn, iters=10, warmup=3):
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="both", choices=["both", "encoder", "decoder"])
    ap.add_argument("--only", default="all", choices=["all", "fused", "round1"],
                    help="time only one launch structure (for a rocprofv3 --kernel-trace --stats run of that structure)")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    B, C, M, L, P = 8, 256, 8, 4, 4
    S = sum(h * w for h, w in CFG4_SHAPES)
    mod = A.MSDeformAttn(C, L, M, P).to(dev).to(torch.bfloat16).eval()
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.01)
    ss = torch.tensor(CFG4_SHAPES, device=dev)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    src = torch.randn(B, S, C, device=dev).to(torch.bfloat16)
    for name, Lq in (("encoder", S), ("decoder", 900)):
        if args.case not in ("both", name):
            continue
        q = torch.randn(B, Lq, C, device=dev).to(torch.bfloat16)
        if Lq == S:   # encoder self-attention: a query's reference point is its own pixel centre (...mask_dn.py:1579-1606)
            pts = []
            for h, w in CFG4_SHAPES:
                ys, xs = torch.meshgrid(torch.arange(h, device=dev) + 0.5, torch.arange(w, device=dev) + 0.5, indexing="ij")
                pts.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
            ref = torch.cat(pts)[None, :, None, :].expand(B, S, L, 2).contiguous()
        else:
            ref = torch.rand(B, Lq, L, 2, device=dev)
        with torch.no_grad():
            from visionllm_amd import _lib
            if args.only != "all":
                _lib.lib().vllm_set_option(b"msda_layer_fused", 1 if args.only == "fused" else 0)
                print(json.dumps(dict(case=name, structure=args.only, us=timeit(lambda: mod(q, ref, src, ss, lsi, None)) * 1e6)))
                continue
            fused = timeit(lambda: mod(q, ref, src, ss, lsi, None))
            _lib.lib().vllm_set_option(b"gemm_skinny", 0)        # round 3: the three linears on the 128 x 128 tile kernel
            fused_tile = timeit(lambda: mod(q, ref, src, ss, lsi, None))
            _lib.lib().vllm_set_option(b"gemm_skinny", 1)
            _lib.lib().vllm_set_option(b"msda_layer_fused", 0)   # round-1 launch structure: 2 query GEMMs + prep + cvt
            fused_r1 = timeit(lambda: mod(q, ref, src, ss, lsi, None))
            _lib.lib().vllm_set_option(b"msda_layer_fused", 1)
            ok = A.msda_layer_fused_ok
            A.msda_layer_fused_ok = lambda *a, **k: False
            try:
                composed = timeit(lambda: mod(q, ref, src, ss, lsi, None))
            finally:
                A.msda_layer_fused_ok = ok
        flops = 2.0 * B * C * (S * C + Lq * (M * L * P * 3) + Lq * C)
        print(json.dumps(dict(case=name, B=B, Lq=Lq, fused_us=fused * 1e6, fused_with_tile_kernel_gemms_us=fused_tile * 1e6, fused_round1_structure_us=fused_r1 * 1e6, composed_us=composed * 1e6,
                              speedup=composed / fused, gemm_gflop=flops / 1e9)))


if __name__ == "__main__":
    main()
