import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
M = 23080
for name, N, K, epi in (("qkv", 3072, 1024, 0), ("proj", 1024, 1024, 0), ("fc1", 4096, 1024, 2), ("fc2", 1024, 4096, 0), ("sq4096", 4096, 4096, 0)):
    Mm = 4096 if name == "sq4096" else M
    x = torch.randn(Mm, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.zeros(N, device="cuda").bfloat16()
    y = torch.empty(Mm, N, device="cuda", dtype=torch.bfloat16)
    ours = timeit(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, epi, None, None, 0, 0, st)))
    _lib.set_option('gemm_direct_store', 1)
    direct = timeit(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, epi, None, None, 0, 0, st)))
    _lib.set_option('gemm_direct_store', 0)
    vialds = timeit(lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), Mm, N, K, K, K, N, epi, None, None, 0, 0, st)))
    _lib.set_option('gemm_direct_store', 2)
    lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * Mm * N * K
    print(json.dumps(dict(shape=name, M=Mm, N=N, K=K, ours_us=ours * 1e6, ours_TF=fl / ours / 1e12, direct_store_us=direct * 1e6, via_lds_us=vialds * 1e6, hipblaslt_us=lib * 1e6, hipblaslt_TF=fl / lib / 1e12)))
