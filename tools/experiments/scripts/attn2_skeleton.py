"""How much of an attention launch is per-BLOCK cost (dispatch, Q / first-tile latency, store drain) and how much per-TILE:
the skeleton-only ablation build of attn2.hip (mask 62: barriers + loop, no DMA after the first tile, no LDS reads, no MFMA,
no softmax) next to the full kernel, same number of blocks (3200), different numbers of KV tiles per block."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = {}
for m in (0, 62):
    L = ctypes.CDLL(os.path.join(ROOT, "visionllm_amd", "_build_abl", f"libattn2_abl{m}.so"))
    L.attn2_abl_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    libs[m] = L
var2 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = torch.cuda.current_stream().cuda_stream
for D, H, cases in ((64, 16, ((200, 65), (100, 193), (40, 577), (20, 1153))), (128, 25, ((72, 129), (24, 385), (8, 1025), (4, 2049)))):
    for n, S in cases:
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
        nqt = (S + 127) // 128
        print(f"d={D} n={n} S={S}: blocks {n * H * nqt}, KV tiles per block {(S + 63) // 64}: full {best[0]:7.1f} us, skeleton only {best[62]:6.1f} us")
