#!/usr/bin/env python3
"""Order-robust A/B of the GEMM dispatch options on the ViT-L shapes: every mode is timed several times in an interleaved
order after a long warm-up and the minimum is reported (a single event-timed loop right after allocation reads 5-10 %
slow: clocks / power state)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib  # noqa: E402

L = _lib.lib()
st = _lib.current_stream()


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    shapes = [("qkv", 23080, 3072, 1024, 0), ("proj", 23080, 1024, 1024, 3), ("fc1", 23080, 4096, 1024, 2),
              ("fc2", 23080, 1024, 4096, 3), ("sq4096", 4096, 4096, 4096, 0)]
    for name, M, N, K, epi in shapes:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        b = torch.zeros(N, device="cuda").bfloat16()
        res = torch.randn(M, N, device="cuda").bfloat16()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

        def run(force, ds):
            def f():
                _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, epi | force,
                                            None, _lib.ptr(res) if epi == 3 else None, N, 0, st))
            _lib.set_option("gemm_direct_store", ds)
            t = timeit(f)
            _lib.set_option("gemm_direct_store", 2)
            return t
        modes = {"default": (0, 2), "256_direct": (0x200, 1), "256_lds": (0x200, 0), "192_direct": (0x300, 1), "192_lds": (0x300, 0), "mf32_lds": (0x800, 0)}
        for _ in range(3):
            for k, (f, d) in modes.items():
                run(f, d)   # warm-up of every code path
        best = {k: 1e9 for k in modes}
        for _ in range(4):
            for k, (f, d) in modes.items():
                best[k] = min(best[k], run(f, d))
        lib = min(timeit(lambda: torch.nn.functional.linear(x, w, b)) for _ in range(4))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(shape=name, **{k: round(v, 1) for k, v in best.items()}, hipblaslt=round(lib, 1),
                              default_TF=round(fl / best["default"] / 1e6, 1))))


if __name__ == "__main__":
    main()
