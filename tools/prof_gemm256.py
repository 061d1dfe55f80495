"""gemm256 phase clock (VLLM_GEMM_PROF=1): prologue / main loop / epilogue ticks of wave 0, averaged over the blocks."""
import ctypes, os, sys, torch
os.environ["VLLM_GEMM_PROF"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
buf = (ctypes.c_long * 16)()
for name, M, N, K, epi in (("qkv", 23080, 3072, 1024, 0), ("proj", 23080, 1024, 1024, 0), ("proj+res", 23080, 1024, 1024, 3), ("fc1", 23080, 4096, 1024, 2),
                           ("fc2", 23080, 1024, 4096, 0), ("fc2+res", 23080, 1024, 4096, 3), ("sq4096", 4096, 4096, 4096, 0), ("ivit_fc1", 8200, 12800, 3200, 1)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda").bfloat16() if epi == 3 else None
    f = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, epi, None, _lib.ptr(res), N if epi == 3 else 0, 0, st))
    for _ in range(3): f()
    L.vllm_debug_counters(buf, 8)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    L.vllm_debug_counters(buf, 8)
    n = max(buf[3], 1)
    print(f"{name:9s} {e0.elapsed_time(e1) * 100:7.1f} us/launch  blocks/launch {n // 10:5d}  ticks per block: prologue {buf[0] / n:7.0f}  "
          f"main loop {buf[1] / n:8.0f} ({buf[1] / n / (K // 64):6.0f} per K tile)  epilogue {buf[2] / n:7.0f}")
