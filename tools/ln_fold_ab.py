#!/usr/bin/env python3
"""Folded-norm consumer GEMM against the plain GEMM, isolated (same box, interleaved rounds).

Separates what the kernel costs from what the DATA costs: the folded consumer multiplies the un-normalised residual stream
(outlier channels, mean != 0) where the plain GEMM multiplies the normalised rows.
  plain(norm)   vllm_gemm_bf16     on the normalised rows            (what the launched-norm path runs)
  plain(raw)    vllm_gemm_bf16     on the un-normalised rows         (same kernel, the other data: wrong numbers, right cost)
  folded(raw)   vllm_gemm_bf16_ln  on the un-normalised rows         (what the folded path runs)
  folded(norm)  vllm_gemm_bf16_ln  on the normalised rows            (folded kernel, the benign data)
ViT-L shapes at batch 64: M = 16448, K = 1024, N = 3072 (qkv, bias) and 4096 (fc1, quick-GELU).
"""
import ctypes
import math
import os
import sys

PHASES = "--phases" in sys.argv   # gemm256's phase clock (ticks of wave 0 per block) instead of launch times
if PHASES:
    os.environ["VLLM_GEMM_PROF"] = "1"

import torch

sys.path.insert(0, ".")
from visionllm_amd import _lib  # noqa: E402

DEV = "cuda:0"


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def bf(t):
    return t.to(torch.bfloat16).contiguous()


def main():
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    M, C = 16448, 1024
    eps = 1e-5
    torch.manual_seed(0)
    raw = torch.randn(M, C, device=DEV) * 2.0 + 0.75
    raw[:, 7] += 40.0
    raw[:, 300] -= 25.0
    raw = bf(raw)
    gamma = bf(1.0 + 0.2 * torch.randn(C, device=DEV))
    beta = bf(0.1 * torch.randn(C, device=DEV))
    rf = raw.float()
    mu = rf.mean(1, keepdim=True)
    nrm = bf((rf - mu) * torch.rsqrt(((rf - mu) ** 2).mean(1, keepdim=True) + eps) * gamma.float() + beta.float())
    # statistics as a producer leaves them: per 256-column tile {mean, M2}
    def stats_of(x):
        xb = x.float().view(M, 4, 256)
        m = xb.mean(2)
        return torch.stack([m, ((xb - m[..., None]) ** 2).sum(2)], dim=2).contiguous()
    st_raw, st_nrm = stats_of(raw), stats_of(nrm)
    out = {}
    for name, N, epi in (("qkv", 3072, 0), ("fc1", 4096, 2)):
        w = bf(torch.randn(N, C, device=DEV) / math.sqrt(C))
        b = bf(torch.randn(N, device=DEV))
        wf = bf(w.float() * gamma.float()[None, :])
        colsum = wf.float().sum(1).contiguous()
        bias_ln = (b.float() + w.float() @ beta.float()).contiguous()
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)

        def plain(x):
            _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, C, C, C, N, epi, None, None, 0, 0, st))

        def folded(x, s):
            _lib.check(L.vllm_gemm_bf16_ln(P(x), P(wf), None, P(y), M, N, C, C, C, N, epi, None, None, 0, None, P(s), 4, 0, eps,
                                           P(colsum), P(bias_ln), st))

        legs = {"plain(norm)": lambda: plain(nrm), "plain(raw)": lambda: plain(raw),
                "folded(raw)": lambda: folded(raw, st_raw), "folded(norm)": lambda: folded(nrm, st_nrm)}
        if PHASES:
            buf = (ctypes.c_long * 16)()
            for k, fn in legs.items():
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                L.vllm_debug_counters(buf, 8)
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                L.vllm_debug_counters(buf, 8)
                n = max(buf[3], 1)
                print(f"{name} {k:13s} blocks/launch {n // 10:5d}  ticks per block: prologue {buf[0] / n:7.0f}  main loop {buf[1] / n:8.0f}  epilogue {buf[2] / n:7.0f}")
            continue
        times = {k: [] for k in legs}
        for rnd in range(4):
            for k, fn in legs.items():
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 40 * 1e3)
        out[name] = {k: round(sorted(v)[len(v) // 2], 1) for k, v in times.items()}
        print(name, "N", N, "us per launch (median of 4 rounds):", out[name], " rounds:", {k: [round(t, 1) for t in v] for k, v in times.items()})
    return out


if __name__ == "__main__":
    main()
