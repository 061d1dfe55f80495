"""Phase clock of the 64-rows-per-wave attention body (attn_variant 2 | 64 | 4096 | 8192): s_memtime ticks of every wave, summed per phase."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
n, S, H, D = 40, 1025, 25, 128
qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
names = ["prologue", "tile top: vmcnt(0) + barrier", "DMA issue", "h0: rescale + K req + exp + QK", "h0: PV + max", "h1: rescale + K req + exp + QK", "h1: PV + max", "epilogue"]
for var in (2 | 64 | 4096 | 8192,):
    _lib.set_option("attn_variant", var)
    buf = (ctypes.c_long * 8)()
    for _ in range(2):
        _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
    torch.cuda.synchronize()
    L.vllm_debug_counters(buf, 8)
    _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
    torch.cuda.synchronize()
    L.vllm_debug_counters(buf, 8)
    tot = sum(buf)
    waves = 1000 * 4 * 4   # pairs x blocks x waves
    print(f"variant {var}: total ticks {tot:.3e}; per wave {tot / waves:.0f}; per wave and 64-key tile {tot / waves / 16:.0f} (s_memtime at 100 MHz: x shader clock / 100 MHz for cycles)")
    for nm, v in zip(names, buf):
        print(f"  {nm:36s} {v / tot * 100:5.1f} %   {v / waves:9.0f} ticks per wave")
_lib.set_option("attn_variant", 32)
