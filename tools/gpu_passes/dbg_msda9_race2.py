"""Generation 9 race: when a run differs, compare the wrong outputs with the per-level contributions (gather kernel with the
attention weights of the other levels zeroed): which level's contribution is wrong?"""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import make_inputs
from test_msda_gpu import PYRAMIDS
from visionllm_amd import _lib, ms_deform_attn as A
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
name = sys.argv[2] if len(sys.argv) > 2 else "L3_mixed_rounding"
shapes = PYRAMIDS[name]
g = make_inputs(2, 8, 32, shapes, 4, mode="encoder_like", seed=len(shapes))
rng = np.random.default_rng(7)
loc = g["loc"].copy(); flat = loc.reshape(-1, 2)
flat[1::3] += rng.standard_normal(flat[1::3].shape).astype(np.float32) * 0.15
flat[3::29] = 1.7; flat[5::97] = np.nan; flat[6::101] = np.inf
g["loc"] = loc
def up(): return {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
t = up()
def run(tt): return A.ms_deform_attn_forward(tt["value"], tt["shapes"], tt["lsi"], tt["loc"], tt["attw"], 64)
_lib.set_option("msda_tiled", 0); ref = run(t)
L = len(shapes)
per_level = []
for l in range(L):
    w = t["attw"].clone(); m = torch.zeros_like(w); m[:, :, :, l] = 1; tt = dict(t); tt["attw"] = w * m
    per_level.append(run(tt))
_lib.set_option("msda_tiled", 20)
starts = np.cumsum([0] + [h * w for h, w in shapes])
shown = 0
for it in range(N):
    o = run(up())
    d = (o - ref).abs()
    bad = torch.nonzero(d.amax(-1) > 1e-4)
    if len(bad) == 0: continue
    print(f"run {it}: {len(bad)} wrong (image, query) pairs", flush=True)
    seen = set()
    for b, q in bad.tolist():
        l = int(np.searchsorted(starts, q, side="right") - 1); r = q - starts[l]; H, W = shapes[l]; y, x = divmod(r, W)
        hd = torch.nonzero(d[b, q].reshape(8, 32).amax(-1) > 1e-4).flatten().tolist()
        for h in hd:
            delta = (o[b, q] - ref[b, q]).reshape(8, 32)[h]
            fits = []
            for ll in range(L):
                c = per_level[ll][b, q].reshape(8, 32)[h]
                fits.append((float((delta + c).abs().max()), float(c.abs().max())))   # delta == -contribution: the level is missing
            key = (l, y >> 3 if l == 0 else -1, x >> 4 if l == 0 else -1, h, b)
            if key in seen: continue
            seen.add(key)
            print(f"   b{b} level {l} (y {y}, x {x}) head {h}: |delta| {float(delta.abs().max()):.3g}; per level (|delta + contribution|, |contribution|): " +
                  " ".join(f"L{ll}: {a:.3g}/{c:.3g}" for ll, (a, c) in enumerate(fits)))
    shown += 1
    if shown >= 6: break
print("done")
