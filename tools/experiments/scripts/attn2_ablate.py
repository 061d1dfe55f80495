"""Times the ablation builds of attn2.hip (tools/attn2_ablate.sh) side by side, interleaved over rounds (same box, same
session): what each part of the tile loop costs.  Results of masks != 0 are wrong by construction."""
import ctypes, glob, os, re, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {0: "full", 1: "no v_exp", 2: "no softmax VALU", 4: "no QK MFMA", 8: "no PV MFMA", 12: "no MFMA", 16: "no DMA",
         32: "no LDS fragment reads", 64: "no barrier", 28: "no MFMA, no DMA", 14: "no MFMA, no softmax", 46: "no MFMA / softmax / LDS reads",
         62: "only barrier + loop", 34: "no softmax, no LDS reads"}
libs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "visionllm_amd", "_build_abl", "libattn2_abl*.so"))):
    m = int(re.search(r"abl(\d+)\.so", f).group(1))
    L = ctypes.CDLL(f)
    L.attn2_abl_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    libs[m] = L
var2 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = torch.cuda.current_stream().cuda_stream
for (n, S, H, D) in ((40, 577, 16, 64), (8, 1025, 25, 128)):
    qkv = torch.randn(n, S, 3, H, D, device="cuda").bfloat16()
    out = torch.empty(n, S, H, D, device="cuda", dtype=torch.bfloat16)
    best = {m: 1e9 for m in libs}
    for _ in range(5):
        for m, L in libs.items():
            f = lambda: L.attn2_abl_run(qkv.data_ptr(), out.data_ptr(), n, S, H, D, D ** -0.5, var2, st)
            for _ in range(3): f()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            best[m] = min(best[m], e0.elapsed_time(e1) / 20 * 1e3)
    for m in sorted(best):
        print(f"d={D} S={S} n={n} var2={var2} mask {m:3d} {NAMES.get(m, '?'):32s} {best[m]:7.1f} us   (full - this = {best[0] - best[m]:6.1f})")
