"""Attention forward timing (ViT-L: 40 tiles x 16 heads x S577 x d64; InternViT-6B: 8 tiles x 25 heads x S1025 x d128):
min over rounds of an event-timed batch of launches, per attn_variant."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()


def run(n, S, H, D, variants=(2,), rounds=7, reps=20):
    qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
    out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    flops = 4.0 * n * H * S * S * D
    best = {v: 1e9 for v in variants}
    for _ in range(rounds):
        for v in variants:
            old = _lib.set_option("attn_variant", v)
            for _ in range(3):
                _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / reps * 1e3)
            _lib.set_option("attn_variant", old)
    for v in variants:
        print(f"attn n={n} S={S} H={H} D={D} variant {v:2d}: {best[v]:7.1f} us  {flops / best[v] / 1e6:6.0f} TFLOP/s "
              f"({flops / best[v] / 1e6 / 2500:.3f} of 2.5 PF)")


if __name__ == "__main__":
    # round 6: 66 = automatic (deferred rescale, O through LDS, class-token split); +128: no split; +1024: class token out of the key
    # tiling only
    NEW = (66, 194, 1090)
    if len(sys.argv) > 1:
        NEW = tuple(int(x) for x in sys.argv[1].split(","))
    run(40, 577, 16, 64, variants=NEW)
    run(40, 1025, 25, 128, variants=NEW)
    run(5, 1025, 25, 128, variants=NEW)
