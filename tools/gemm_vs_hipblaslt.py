"""gemm256 (ours) against hipBLASLt (through torch.nn.functional.linear) on the encoder GEMM shapes, LIKE FOR LIKE (bias-only epilogue
on both sides) and SUSTAINED (each measurement loops for ~1.5 s: under dense MFMA work the part settles at ~2.0-2.1 GHz at the package
power limit, short bursts right after an idle gap read differently).  Also: ours with the epilogue the encoder actually uses."""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()


def sustained(fn, secs=1.5):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); it = 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        it += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


M = 23080
for name, Mm, N, K, epi, epiname in (("qkv", M, 3072, 1024, 0, "bias"), ("proj", M, 1024, 1024, 3, "bias+residual (CLIP: no LayerScale)"),
                                     ("fc1", M, 4096, 1024, 2, "bias+quick_gelu"), ("fc2", M, 1024, 4096, 3, "bias+residual (CLIP: no LayerScale)"),
                                     ("sq4096", 4096, 4096, 4096, 0, "bias"), ("ivit_fc1", 8200, 12800, 3200, 1, "bias+gelu(erf)")):
    x = torch.randn(Mm, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.zeros(N, device="cuda").bfloat16()
    y = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(Mm, N, device="cuda").bfloat16(); ls = torch.ones(N, device="cuda").bfloat16()
    ours_bias = sustained(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, 0, None, None, 0, 0, st)))
    lib = sustained(lambda: torch.nn.functional.linear(x, w, b))
    if epi == 3:
        ours_epi = sustained(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, 3, None, _lib.ptr(res), N, 0, st)))
        ours_ls = sustained(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, 3, _lib.ptr(ls), _lib.ptr(res), N, 0, st)))
    elif epi:
        ours_epi = sustained(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, epi, None, None, 0, 0, st)))
    else:
        ours_epi = ours_bias
    extra = dict(ours_with_layerscale_residual_us=round(ours_ls, 1)) if epi == 3 else {}
    fl = 2.0 * Mm * N * K
    print(json.dumps(dict(shape=name, M=Mm, N=N, K=K, ours_bias_us=round(ours_bias, 1), hipblaslt_bias_us=round(lib, 1),
                          ours_over_hipblaslt=round(lib / ours_bias, 3), ours_bias_TF=round(fl / ours_bias / 1e6, 0), hipblaslt_TF=round(fl / lib / 1e6, 0),
                          ours_with_encoder_epilogue_us=round(ours_epi, 1), encoder_epilogue=epiname, **extra)))
