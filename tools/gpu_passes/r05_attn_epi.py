"""round 5: attention epilogue A/B -- attn_variant 2 (8-byte stores from the accumulator layout) vs 66 (O through LDS, whole-row 16-byte
stores); outputs must be bit-identical (same values, same rounding, only the path to memory differs)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from visionllm_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = _lib.lib(); st = _lib.current_stream()
for (n, S, H, D) in ((3, 577, 16, 64), (2, 1025, 25, 128), (2, 130, 4, 64), (1, 64, 2, 128), (2, 257, 3, 64)):
    for dt, fn in ((torch.bfloat16, L.vllm_attn_fwd_qkvpacked_bf16), (torch.float16, L.vllm_attn_fwd_qkvpacked_f16)):
        qkv = torch.randn(n, S, 3, H, D, device="cuda").to(dt)
        outs = []
        for v in (2, 66):
            _lib.set_option("attn_variant", v)
            o = torch.full((n, S, H, D), float("nan"), device="cuda", dtype=dt)
            _lib.check(fn(_lib.ptr(qkv), _lib.ptr(o), n, S, H, D, D ** -0.5, st))
            outs.append(o)
        print((n, S, H, D), dt, "bit-identical:", torch.equal(outs[0], outs[1]), "finite:", bool(torch.isfinite(outs[1].float()).all()), flush=True)
_lib.set_option("attn_variant", 32)
import importlib.util
spec = importlib.util.spec_from_file_location("bench_attn", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bench_attn.py"))
ba = importlib.util.module_from_spec(spec); spec.loader.exec_module(ba)
ba.run(40, 577, 16, 64, variants=(2, 66))
ba.run(8, 1025, 25, 128, variants=(2, 66))
ba.run(40, 1025, 25, 128, variants=(2, 66))
