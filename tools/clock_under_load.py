"""Shader clock / package power sampled with rocm-smi while one kernel runs in a tight loop (is the part power- or
current-limited under this kernel?).  usage: clock_under_load.py attn|attn_zeros|attn_small|gemm|gemm_zeros|msda|dcnv3|idle [seconds]"""
import os, subprocess, sys, threading, time, re
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0

if what.startswith("attn"):
    n, S, H, D = 40, 577, 16, 64
    qkv = (torch.zeros(n, S, 3, H, D, device="cuda") if what == "attn_zeros" else
           torch.randn(n, S, 3, H, D, device="cuda") * (1e-3 if what == "attn_small" else 1.0)).bfloat16(); out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    fn = lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(out), n, S, H, D, D ** -0.5, st))
elif what in ("gemm", "gemm_zeros"):
    M, N, K = 23080, 4096, 1024
    z = 0.0 if what == "gemm_zeros" else 1.0
    x = (torch.randn(M, K, device="cuda") * z).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02 * z).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, 2, None, None, 0, 0, st))
elif what == "hipblaslt":   # the same fc1 shape through torch (hipBLASLt), bias only
    M, N, K = 23080, 4096, 1024
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16()
    fn = lambda: torch.nn.functional.linear(x, w, b)
elif what == "gemm_bias":   # ours, bias only (the like-for-like of hipblaslt)
    M, N, K = 23080, 4096, 1024
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, 0, None, None, 0, 0, st))
elif what == "msda":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    t = bench.build_msda_inputs("cuda", 8, 1)["enc"]
    from visionllm_amd import ms_deform_attn as A
    fn = lambda: A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
elif what == "dcnv3":
    from visionllm_amd import dcnv3 as DC
    N_, H_, W_, G_, C_, k_ = 8, 168, 168, 20, 32, 3
    xi = torch.randn(N_, H_, W_, G_ * C_, device="cuda"); of = torch.randn(N_, H_, W_, G_ * k_ * k_ * 2, device="cuda")
    mk = torch.softmax(torch.randn(N_, H_, W_, G_, k_ * k_, device="cuda"), -1).reshape(N_, H_, W_, -1)
    fn = lambda: DC.dcnv3_forward(xi, of, mk, k_, k_, 1, 1, 1, 1, 1, 1, G_, C_, 1.0)
else:
    fn = lambda: time.sleep(0.001)

samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
            samples.append((sclk.group(1) if sclk else "?", pw.group(1) if pw else "?"))
        except Exception as e:
            samples.append(("err", str(e)[:40]))
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); it = 0
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    it += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print(what, "iters", it, "avg us", round(e0.elapsed_time(e1) * 1e3 / max(it, 1), 1), "samples (sclk MHz, W):", samples)
