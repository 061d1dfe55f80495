"""Debug: DCNv3 backward windowed kernel vs the fp32 C oracle on one case; prints the worst violations per output."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dcnv3 as O
from visionllm_amd import dcnv3 as A, _lib
N, H, W, G, C, k, s, p, d, scale = [float(x) if i == 9 else int(x) for i, x in enumerate(sys.argv[1:11])]
rng = np.random.default_rng(N * 100 + H + C)
Ho, Wo = O.out_size(H, W, k, k, s, s, p, p, d, d)
inp = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
off = (rng.standard_normal((N, Ho, Wo, G * k * k * 2)) * 2.5).astype(np.float32)
if os.environ.get("NAN", "1") == "1":
    off.reshape(-1)[5::97] = np.nan
    off.reshape(-1)[11::131] = np.inf
msk = rng.random((N, Ho, Wo, G * k * k)).astype(np.float32)
go = rng.standard_normal((N, Ho, Wo, G * C)).astype(np.float32)
tt = lambda a: torch.from_numpy(a).to("cuda:0")
ri, ro, rm = O.backward(inp, off, msk, go, k, k, s, s, p, p, d, d, G, C, scale)
mi, _, _ = O.backward(np.abs(inp), off, np.abs(msk), np.abs(go), k, k, s, s, p, p, d, d, G, C, scale)
for v in (0, 1):
    _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", v)
    gi, gof, gm = A.dcnv3_backward(tt(inp), tt(off), tt(msk), k, k, s, s, p, p, d, d, G, C, scale, tt(go))
    e = np.abs(gi.cpu().numpy() - ri)
    ratio = e / (2.0 ** -17 * mi + 1e-7)
    idx = np.unravel_index(np.argmax(ratio), ratio.shape)
    print("tiled", v, "grad_input worst ratio %.2f at %s: got %.6g ref %.6g mag %.4g; #viol %d; max abs err %.3g" % (
        ratio.max(), idx, gi.cpu().numpy()[idx], ri[idx], mi[idx], int((ratio > 1).sum()), e.max()))
    print("   grad_offset max err %.3g (scale %.3g)  grad_mask max err %.3g (scale %.3g)" % (
        np.abs(gof.cpu().numpy() - ro).max(), np.abs(ro).max(), np.abs(gm.cpu().numpy() - rm).max(), np.abs(rm).max()))
