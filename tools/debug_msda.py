import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from msda_inputs import CFG4_SHAPES, make_inputs
from oracle import msda as O
from visionllm_amd import _lib, ms_deform_attn as A
g = make_inputs(2, 8, 32, CFG4_SHAPES, 4, mode="encoder_like", seed=1)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
_lib.set_option("msda_tiled", 0)
b, q, m = 0, 8363, 5
shapes = g["shapes"]
for j in range(16):
    aw = np.zeros_like(g["attw"]); aw.reshape(2, -1, 8, 16)[:, :, :, j] = 1.0
    out = A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], torch.from_numpy(aw).cuda(), 64).cpu().numpy()
    ref = O.forward(g["value"][b:b+1], shapes, g["lsi"], g["loc"][b:b+1, q:q+1], aw[b:b+1, q:q+1])
    d = np.abs(out[b, q, m*32:(m+1)*32] - ref[0, 0, m*32:(m+1)*32]).max()
    l, p = j // 4, j % 4
    x, y = g["loc"][b, q, m, l, p]
    H, W = shapes[l]
    hi = np.float32(np.float32(y) * np.float32(H)) - np.float32(0.5); wi = np.float32(np.float32(x) * np.float32(W)) - np.float32(0.5)
    print(f"point {j} (l{l},p{p}) maxdiff {d:.3e}  h_im {hi!r} w_im {wi!r}")
