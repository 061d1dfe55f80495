#!/usr/bin/env python3
"""Persistent 8-phase GEMM (gemm256p.hip) against one workgroup per tile (VLLM_GEMM_FORCE_TILEWISE), same process, interleaved
rounds; ViT-L shapes at the bench batch (M = 23080).  --phases: the kernels' clocks (ticks of wave 0) instead of launch times."""
import ctypes
import math
import os
import sys

PHASES = "--phases" in sys.argv
if PHASES:
    os.environ.setdefault("VLLM_GEMM_PROF", "1")   # (3: the persistent kernel without its stores -- ablation)

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib  # noqa: E402

L = _lib.lib()
st = _lib.current_stream()
P = _lib.ptr
M = 23080
for name, N, K, epi in (("qkv", 3072, 1024, 0), ("fc1", 4096, 1024, 2), ("proj", 1024, 1024, 3), ("fc2", 1024, 4096, 3), ("ivit_qkv", 9600, 3200, 0), ("ivit_fc1", 12800, 3200, 1),
                         ("ivit_fc2", 3200, 12800, 3)):
    Mx = M if K == 1024 else 8200
    x = torch.randn(Mx, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(Mx, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(Mx, N, device="cuda").bfloat16() if epi == 3 else None
    legs = {"persistent": 0, "tilewise": 0x1000}
    if PHASES:
        buf = (ctypes.c_long * 16)()
        for k, fl in legs.items():
            f = lambda: _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), Mx, N, K, K, K, N, epi | fl, None, P(res) if epi == 3 else None, N if epi == 3 else 0, 0, st))  # noqa: E731
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            L.vllm_debug_counters(buf, 12)
            for _ in range(10):
                f()
            torch.cuda.synchronize()
            L.vllm_debug_counters(buf, 12)
            if buf[6]:
                print(f"{name:9s} {k:10s} tiles/launch {buf[6] // 10:5d}  ticks per tile: main loop {buf[4] / buf[6]:8.0f} (K tile 0 / 1 / 2 / 3: {buf[7] / buf[6]:5.0f} {buf[8] / buf[6]:5.0f} {buf[9] / buf[6]:5.0f} {buf[10] / buf[6]:5.0f})  epilogue {buf[5] / buf[6]:7.0f}")
            else:
                n = max(buf[3], 1)
                print(f"{name:9s} {k:10s} blocks/launch {n // 10:5d}  ticks per block: prologue {buf[0] / n:7.0f}  main loop {buf[1] / n:8.0f}  epilogue {buf[2] / n:7.0f}")
        continue
    times = {k: [] for k in legs}
    for rnd in range(4):
        for k, fl in legs.items():
            f = lambda: _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), Mx, N, K, K, K, N, epi | fl, None, P(res) if epi == 3 else None, N if epi == 3 else 0, 0, st))  # noqa: E731
            for _ in range(5):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 40 * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    fl = 2.0 * Mx * N * K
    print(f"{name:9s} M{Mx} N{N} K{K}: persistent {med['persistent']:7.1f} us ({fl / med['persistent'] / 1e6 / 2500:.3f} of bf16 peak)   "
          f"tilewise {med['tilewise']:7.1f} us ({fl / med['tilewise'] / 1e6 / 2500:.3f})   rounds {[round(t, 1) for t in times['persistent']]} / {[round(t, 1) for t in times['tilewise']]}")
