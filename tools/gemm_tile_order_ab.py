#!/usr/bin/env python3
"""Persistent GEMM: dense XCD tile order (0) against the banded order with RB row panels per band (vllm_set_option("gemm_tile_rb")),
same process, interleaved rounds; bit-identity of the outputs checked.  ViT-L shapes at M = 23080, InternViT-6B shapes at M = 41000."""
import math, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream(); P = _lib.ptr
RBS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,2,4,8,16".split(","))]
for name, M, N, K, epi in (("qkv", 23080, 3072, 1024, 0), ("fc1", 23080, 4096, 1024, 2), ("proj", 23080, 1024, 1024, 3), ("fc2", 23080, 1024, 4096, 3),
                           ("ivit_qkv", 41000, 9600, 3200, 0), ("ivit_fc1", 41000, 12800, 3200, 1), ("ivit_proj", 41000, 3200, 3200, 3), ("ivit_fc2", 41000, 3200, 12800, 3)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16() if epi == 3 else None
    f = lambda: _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, P(res) if epi == 3 else None, N if epi == 3 else 0, 0, st))
    outs, times = {}, {rb: [] for rb in RBS}
    for rb in RBS:
        _lib.set_option("gemm_tile_rb", rb); y.zero_(); f(); torch.cuda.synchronize(); outs[rb] = y.clone()
    same = all(torch.equal(outs[RBS[0]], outs[rb]) for rb in RBS)
    iters = 40 if K * N < 2e7 else 12
    for rnd in range(4):
        for rb in RBS:
            _lib.set_option("gemm_tile_rb", rb)
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): f()
            e1.record(); torch.cuda.synchronize()
            times[rb].append(e0.elapsed_time(e1) / iters * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:10s} M{M} N{N} K{K}: bit-identical {same}  " + "  ".join(f"rb{rb}: {min(times[rb]):7.1f} us ({fl / min(times[rb]) / 2.5e9:.3f})" for rb in RBS), flush=True)
_lib.set_option("gemm_tile_rb", -1)
