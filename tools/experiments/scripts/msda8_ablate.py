"""Times the ablation builds of msda_tiled8.hip (tools/msda8_ablate.sh) side by side at BASELINE cfg 4 (B = 8, encoder shape),
interleaved over rounds: what each part of the pyramid-item pipeline costs.  Results of masks != 0 are wrong by construction."""
import ctypes, glob, os, re, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
NAMES = {32: "no point arithmetic (P1 skipped)", 9000: "HEAD before this change (round-2 DMA rounds)", 1024: "round-2 placement: DMA round BETWEEN gather points", 1028: "round-2 placement, no window DMA", 128: "DMA addresses computed, no load issued", 256: "DMA from the zero line only", 0: "full", 1: "no gather FMAs", 2: "no gather LDS reads", 3: "no gather reads + FMAs (addressing / broadcasts kept)", 4: "no window DMA",
         8: "no stores", 16: "no gather", 20: "no gather, no DMA", 28: "no gather / DMA / stores", 64: "no barrier", 6: "no reads, no DMA",
         92: "no gather / DMA / stores / barrier"}
libs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "visionllm_amd", "_build_abl", "libmsda8_abl*.so"))):
    m = int(re.search(r"abl(\d+)\.so", f).group(1))
    L = ctypes.CDLL(f)
    L.t8_abl_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
    libs[m] = L
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "encoder_like"
g = make_inputs(1, 8, 32, CFG4_SHAPES, 4, mode=mode, seed=0)
t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
B = 8
for k in ("value", "loc", "attw"):
    t[k] = t[k].repeat(B, *([1] * (t[k].dim() - 1))).contiguous()
t["value"] = t["value"] + 0.01 * torch.randn_like(t["value"])
_, S, M, D = t["value"].shape
Lq, Lv = t["loc"].shape[1], t["loc"].shape[3]
out = torch.empty(B, Lq, M * D, device=dev)
st = torch.cuda.current_stream().cuda_stream
best = {m: 1e9 for m in libs}
for _ in range(5):
    for m, L in libs.items():
        f = lambda: L.t8_abl_run(t["value"].data_ptr(), t["shapes"].data_ptr(), t["lsi"].data_ptr(), t["loc"].data_ptr(), t["attw"].data_ptr(),
                                 B, S, M, Lv, Lq, out.data_ptr(), st)
        for _ in range(2): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best[m] = min(best[m], e0.elapsed_time(e1) / 10 * 1e3)
for m in sorted(best):
    print(f"{mode} mask {m:3d} {NAMES.get(m, '?'):55s} {best[m]:7.1f} us   (full - this = {best[0] - best[m]:6.1f})")
