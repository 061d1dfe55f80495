"""Race screen for generation 9 (msda_tiled 20): the 'mixed' location set on every PYRAMIDS geometry, N runs each against the
gather kernel; prints where mismatching outputs sit (image, level, y, x, head) so that the slot / pass / team can be read off."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import make_inputs
from test_msda_gpu import PYRAMIDS
from visionllm_amd import _lib, ms_deform_attn as A
import ctypes
dev = "cuda:0"
SIDE = os.environ.get("MSDA9_LIB")     # a side build of visionllm_amd/csrc/msda_tiled9.hip (tools/msda9_variants.sh) instead of a library mode
side = None
if SIDE:
    side = ctypes.CDLL(SIDE)
    side.t9_abl_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode_id = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name in sorted(PYRAMIDS):
    shapes = PYRAMIDS[name]
    for mode in ("mixed", "encoder_like", "uniform"):
        g = make_inputs(2, 8, 32, shapes, 4, mode="encoder_like", seed=len(shapes))
        rng = np.random.default_rng(7)
        if mode == "mixed":
            loc = g["loc"].copy(); flat = loc.reshape(-1, 2)
            flat[1::3] += rng.standard_normal(flat[1::3].shape).astype(np.float32) * 0.15
            flat[3::29] = 1.7; flat[5::97] = np.nan; flat[6::101] = np.inf
            g["loc"] = loc
        elif mode == "uniform":
            g["loc"] = (rng.random(g["loc"].shape, dtype=np.float32) * 1.1 - 0.05).astype(np.float32)
        fresh = os.environ.get("FRESH", "1") == "1"
        use_side = [False]
        t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
        def run():
            tt = {k: torch.from_numpy(v).to(dev) for k, v in g.items()} if fresh else t     # (as the tests do: new tensors, cold caches)
            if side is not None and use_side[0]:
                Bv, Sv, Mv, Dv = tt["value"].shape
                o = torch.empty(Bv, tt["loc"].shape[1], Mv * Dv, device=dev)
                rc = side.t9_abl_run(tt["value"].data_ptr(), tt["shapes"].data_ptr(), tt["lsi"].data_ptr(), tt["loc"].data_ptr(), tt["attw"].data_ptr(),
                                     Bv, Sv, Mv, tt["loc"].shape[3], tt["loc"].shape[1], o.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0
                return o
            return A.ms_deform_attn_forward(tt["value"], tt["shapes"], tt["lsi"], tt["loc"], tt["attw"], 64)
        use_side = [False]
        _lib.set_option("msda_tiled", 0); ref = run()
        _lib.set_option("msda_tiled", mode_id)
        use_side[0] = True
        first = run(); bad = 0; worst = 0.0; where = {}
        for it in range(N):
            o = run()
            if not torch.equal(o, first):
                bad += 1
                d = (o - first).abs().amax(-1)          # [B, Lq] over the M*D outputs of a query
                idx = torch.nonzero(d > 0)
                worst = max(worst, float((o - ref).abs().max()), float((first - ref).abs().max()))
                starts = np.cumsum([0] + [h * w for h, w in shapes])
                for b, q in idx[:2000].tolist():
                    l = int(np.searchsorted(starts, q, side="right") - 1); r = q - starts[l]; H, W = shapes[l]
                    y, x = divmod(r, W)
                    heads = torch.nonzero((o[b, q] - first[b, q]).reshape(8, 32).abs().amax(-1) > 0).flatten().tolist()
                    key = (l, (y << l) // 8, (x << l) // 16)
                    where.setdefault(key, set()).add((b, tuple(heads), y, x))
        print(f"{name:20s} {mode:12s} runs differing from the first: {bad}/{N}  max |err| vs gather kernel {float((first - ref).abs().max()):.3g} (bad runs {worst:.3g})", flush=True)
        for key in sorted(where)[:12]:
            v = sorted(where[key])[:6]
            print("      level %d item tile (%d,%d): %s" % (key[0], key[1], key[2], v))
