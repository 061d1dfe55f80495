"""gemm256 block timeline (VLLM_GEMM_TRACE): per block {start, end} in 100 MHz s_memrealtime ticks + HW_ID -> per CU: busy time
inside blocks, gaps between consecutive blocks (block turnover: drain of the old block's stores, LDS / register release, launch
of the next 8 waves), first start / last end against the kernel's duration.
The trace path is compiled only into a debug build of the library:
    make -C visionllm_amd/csrc OUT=../_build_trace EXTRA=-DVLLM_GEMM_TRACE_ENABLE -j8
    VLLM_HIP_LIB=$PWD/visionllm_amd/_build_trace/libvllm_hip.so python tools/trace_gemm256.py"""
import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
trace = torch.zeros(8192 * 3, dtype=torch.int64, device="cuda")
os.environ["VLLM_GEMM_TRACE"] = hex(trace.data_ptr())
from visionllm_amd import _lib
L = _lib.lib(); st = _lib.current_stream()
for name, M, N, K, epi in (("qkv", 23080, 3072, 1024, 0), ("fc1", 23080, 4096, 1024, 2), ("fc2", 23080, 1024, 4096, 0),
                           ("sq4096", 4096, 4096, 4096, 0)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda").bfloat16(); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K, K, K, N, epi, None, None, 0, 0, st))
    for _ in range(3): f()
    torch.cuda.synchronize(); trace.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(); f(); e1.record(); torch.cuda.synchronize()
    t = trace.view(-1, 3).cpu().numpy()
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    percu = collections.defaultdict(list)
    full = trace.view(-1, 3).cpu().numpy()
    for bi in range(full.shape[0]):
        s_, e_, hw = full[bi]
        if s_ == 0:
            continue
        # HW_ID: cu_id bits 8..11, sh_id 12, se_id 13..15; the XCD is not in it: blocks are dealt round-robin, XCD = blockIdx % 8
        percu[(int(hw) & 0xff00, bi % 8)].append((int(s_ - t0), int(e_ - t0)))
    dur = (t[:, 1] - t[:, 0]) / 100.0
    gaps, first, last = [], [], []
    for cu, v in percu.items():
        v.sort()
        first.append(v[0][0] / 100.0); last.append(v[-1][1] / 100.0)
        gaps += [(v[i + 1][0] - v[i][1]) / 100.0 for i in range(len(v) - 1)]
    import numpy as np
    print(f"{name:7s} kernel {e0.elapsed_time(e1) * 1e3:7.1f} us  blocks {len(t):5d}  distinct HW_ID keys {len(percu):4d}  block {np.median(dur):6.2f} us (p10 {np.percentile(dur, 10):.2f}"
          f" p90 {np.percentile(dur, 90):.2f})  gap between blocks of a CU: median {np.median(gaps) if gaps else 0:5.2f} us mean {np.mean(gaps) if gaps else 0:5.2f}"
          f"  first start {np.median(first):5.2f} us  last end {max(last):7.2f} us")
