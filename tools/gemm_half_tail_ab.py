#!/usr/bin/env python3
"""Persistent 8-phase GEMM: whole tiles in the last, incomplete round (0) against HALF-HEIGHT tiles there (1: each tile of the last round
as two half tiles on two blocks when they still fit the grid; vllm_set_option("gemm_half_tail")), same process, interleaved rounds;
bit-identity checked.
ViT-L shapes at M = 23080 (40 tiles) and 18464 (32 tiles), InternViT-6B shapes at M = 41000."""
import math, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream(); P = _lib.ptr
for name, M, N, K, epi in (("qkv", 23080, 3072, 1024, 0), ("fc1", 23080, 4096, 1024, 2), ("proj", 23080, 1024, 1024, 3), ("fc2", 23080, 1024, 4096, 3),
                           ("bridge0", 23040, 4096, 1024, 1), ("bridge1", 23040, 4096, 4096, 0),
                           ("qkv32", 18464, 3072, 1024, 0), ("fc1_32", 18464, 4096, 1024, 2), ("proj32", 18464, 1024, 1024, 3), ("fc2_32", 18464, 1024, 4096, 3),
                           ("ivit_qkv", 41000, 9600, 3200, 0), ("ivit_fc1", 41000, 12800, 3200, 1), ("ivit_proj", 41000, 3200, 3200, 3), ("ivit_fc2", 41000, 3200, 12800, 3)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16() if epi == 3 else None
    f = lambda: _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, P(res) if epi == 3 else None, N if epi == 3 else 0, 0, st))
    outs, times, cut = {}, {0: [], 1: []}, {}
    for v in (0, 1):
        _lib.set_option("gemm_half_tail", v); y.zero_(); n0 = L.vllm_gemm_half_tail_launches(); f(); torch.cuda.synchronize(); outs[v] = y.clone()
        cut[v] = L.vllm_gemm_half_tail_launches() - n0
    same = torch.equal(outs[0], outs[1])
    iters = 40 if K * N < 2e7 else 12
    for rnd in range(4):
        for v in (0, 1):
            _lib.set_option("gemm_half_tail", v)
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / iters * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:10s} M{M} N{N} K{K}: bit-identical {same}  one launch {min(times[0]):7.1f} us ({fl / min(times[0]) / 2.5e9:.3f})   "
          f"half tail {'(taken)' if cut[1] else '(not taken)':11s} {min(times[1]):7.1f} us ({fl / min(times[1]) / 2.5e9:.3f})", flush=True)
_lib.set_option("gemm_half_tail", 1)
