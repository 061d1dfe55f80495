#!/usr/bin/env python3
"""Per-kernel micro-benchmarks (GPU box): ours vs the vendor library reached through torch (hipBLASLt / SDPA) as a
known-good ceiling on the same hardware, same random data.  Prints one JSON line per measurement."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionllm_amd import _lib  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
P = _lib.ptr


def t(fn, iters=20, warm=3):
    for _ in range(warm):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,attn,norm")
    ap.add_argument("--tiles", type=int, default=40)
    ap.add_argument("--no-ref", action="store_true")
    a = ap.parse_args()
    st = _lib.current_stream(torch.device(dev))
    what = a.what.split(",")
    if "gemm" in what:
        shapes = [("vitl_qkv", a.tiles * 577, 3072, 1024, 0), ("vitl_proj", a.tiles * 577, 1024, 1024, 3),
                  ("vitl_fc1", a.tiles * 577, 4096, 1024, 2), ("vitl_fc2", a.tiles * 577, 1024, 4096, 3),
                  ("ivit_qkv", 5 * 1025, 9600, 3200, 0), ("ivit_fc1", 5 * 1025, 12800, 3200, 1),
                  ("ivit_fc2", 5 * 1025, 3200, 12800, 3), ("sq4096", 4096, 4096, 4096, 0)]
        for name, M, N, K, epi in shapes:
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            b = torch.randn(N, device=dev).to(torch.bfloat16)
            res = torch.randn(M, N, device=dev).to(torch.bfloat16)
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            f = lambda: _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None,  # noqa: E731
                                                     P(res) if epi == 3 else None, N, 0, st))
            sec = t(f)
            fl = 2.0 * M * N * K
            r = dict(kernel="gemm", name=name, M=M, N=N, K=K, epi=epi, us=sec * 1e6, TF=fl / sec / 1e12)
            if not a.no_ref:
                sec2 = t(lambda: torch.nn.functional.linear(x, w, b))
                r["torch_linear_us"] = sec2 * 1e6
                r["torch_linear_TF"] = fl / sec2 / 1e12
            print(json.dumps(r), flush=True)
    if "attnsweep" in what:
        for name, B, S, H, D in [("vitl", a.tiles, 577, 16, 64), ("ivit6b_40", 40, 1025, 25, 128)]:
            qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
            out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev)
            f = lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(P(qkv), P(out), B, S, H, D, D ** -0.5, st))  # noqa: E731
            fl = 4.0 * B * H * S * S * D
            for rnd in range(2):
                for var in (0, 2, 6, 8, 10, 14):
                    _lib.set_option("attn_variant", var)
                    sec = t(f, iters=10)
                    print(json.dumps(dict(kernel="attn", name=name, variant=var, round=rnd, us=sec * 1e6, TF=fl / sec / 1e12)), flush=True)
    if "attn" in what:
        for name, B, S, H, D in [("vitl", a.tiles, 577, 16, 64), ("ivit6b", 5, 1025, 25, 128), ("ivit6b_40", 40, 1025, 25, 128)]:
            qkv = torch.randn(B, S, 3, H, D, device=dev).to(torch.bfloat16)
            out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev)
            f = lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(P(qkv), P(out), B, S, H, D, D ** -0.5, st))  # noqa: E731
            sec = t(f)
            fl = 4.0 * B * H * S * S * D
            r = dict(kernel="attn", name=name, B=B, S=S, H=H, D=D, us=sec * 1e6, TF=fl / sec / 1e12)
            if not a.no_ref:
                q, k, v = [z.permute(0, 2, 1, 3).contiguous() for z in qkv.unbind(2)]
                try:
                    sec2 = t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
                    r["torch_sdpa_us"] = sec2 * 1e6
                    r["torch_sdpa_TF"] = fl / sec2 / 1e12
                except Exception as e:  # noqa: BLE001
                    r["torch_sdpa_err"] = str(e)[:80]
            print(json.dumps(r), flush=True)
    if "norm" in what:
        for name, rows, C, rms in [("vitl_ln", a.tiles * 577, 1024, False), ("ivit_rms", 5 * 1025, 3200, True),
                                   ("ivit_rms40", 40 * 1025, 3200, True)]:
            x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
            w = torch.ones(C, device=dev).to(torch.bfloat16)
            bb = torch.zeros(C, device=dev).to(torch.bfloat16)
            y = torch.empty_like(x)
            if rms:
                f = lambda: _lib.check(L.vllm_rmsnorm_bf16(P(x), C, P(w), P(y), C, rows, C, 1e-6, st))  # noqa: E731
            else:
                f = lambda: _lib.check(L.vllm_layernorm_bf16(P(x), C, P(w), P(bb), P(y), C, rows, C, 1e-5, st))  # noqa: E731
            sec = t(f)
            print(json.dumps(dict(kernel="norm", name=name, rows=rows, C=C, us=sec * 1e6,
                                  GBs=2.0 * rows * C * 2 / sec / 1e9)), flush=True)


if __name__ == "__main__":
    main()
