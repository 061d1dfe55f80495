#!/usr/bin/env python3
"""MSDA forward micro-benchmark at BASELINE cfg 4 (run on the GPU box)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs  # noqa: E402
from visionllm_amd import ms_deform_attn as A  # noqa: E402


def algorithmic_bytes(B, S, M, D, L, Lq, P, vbytes=4):
    # value + loc + attw + out
    return B * S * M * D * vbytes + B * Lq * M * L * P * 2 * 4 + B * Lq * M * L * P * 4 + B * Lq * M * D * vbytes


def timeit(fn, iters, warmup=3, rounds=3):
    """min over `rounds` event-timed loops (the first loop after an idle gap / on freshly allocated operands reads 5-12 % slow:
    clocks ramp, first-touch TLB fills -- a single loop made the first variant of a list look slower than the same kernel later)."""
    best = 1e9
    for _ in range(rounds):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="encoder_like,stress")
    a = ap.parse_args()
    dev = "cuda:0"
    res = []
    for mode in a.modes.split(","):
        for lq in (None, 900):
            g1 = make_inputs(1, 8, 32, CFG4_SHAPES, 4, Lq=lq if (lq or mode != "stress") else 37485, mode=mode, seed=0)
            t = {k: torch.from_numpy(v).to(dev) for k, v in g1.items()}
            for k in ("value", "loc", "attw"):  # replicate batch on device (same distribution, distinct memory)
                t[k] = t[k].repeat(a.B, *([1] * (t[k].dim() - 1))).contiguous()
                if k == "value":
                    t[k] = t[k] + 0.01 * torch.randn_like(t[k])
            B, S, M, D = t["value"].shape
            Lq, L, P = t["loc"].shape[1], t["loc"].shape[3], t["loc"].shape[4]
            for dt in ("f32_auto", "f32_gen6", "f32_gen4", "f32_gen2", "f32_gather", "bf16_auto", "bf16_gather"):
                v = t["value"] if not dt.startswith("bf16") else t["value"].bfloat16()
                from visionllm_amd import _lib
                # f32_auto: generation 7 on pyramids (msda_tiled7.hip), generation 4 otherwise; bf16_auto: generation 6 / gather kernel
                _lib.set_option("msda_tiled", {"f32_gather": 0, "bf16_gather": 0, "f32_auto": 1, "f32_gen4_w8": 2, "f32_gen2": 3,
                                               "f32_gen4_560": 8, "f32_gen4": 9, "f32_gen6": 17}.get(dt, 1))
                sec = timeit(lambda: A.ms_deform_attn_forward(v, t["shapes"], t["lsi"], t["loc"], t["attw"], 64), a.iters)
                ab = algorithmic_bytes(B, S, M, D, L, Lq, P, 2 if dt.startswith("bf16") else 4)
                gathered = B * Lq * M * L * P * 4 * D * (2 if dt.startswith("bf16") else 4)
                r = dict(mode=mode, dtype=dt, B=B, Lq=Lq, us=sec * 1e6, algo_GBs=ab / sec / 1e9,
                         frac_of_8TBs=ab / sec / 8e12, gathered_TBs=gathered / sec / 1e12)
                res.append(r)
                print(json.dumps(r))
    # backward (training of the det heads): encoder shape, fp32
    g1 = make_inputs(1, 8, 32, CFG4_SHAPES, 4, mode="encoder_like", seed=0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in g1.items()}
    for k in ("value", "loc", "attw"):
        t[k] = t[k].repeat(a.B, *([1] * (t[k].dim() - 1))).contiguous()
    go = torch.randn(a.B, t["loc"].shape[1], 256, device=dev)
    from visionllm_amd import _lib
    for name, mode in (("f32_backward_tiled", 1), ("f32_backward_plain_atomics", 0)):
        old = _lib.set_option("msda_tiled", mode)
        sec = timeit(lambda: A.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], go, 64), 5)
        _lib.set_option("msda_tiled", old)
        print(json.dumps(dict(mode="encoder_like", dtype=name, B=a.B, Lq=int(t["loc"].shape[1]), us=sec * 1e6)))
    return res


if __name__ == "__main__":
    main()
