"""Region-encoder point sampling on the GPU (libvllm_hip.so through the C ABI) against the oracle / reference golden."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import region as O
from visionllm_amd import region_encoder as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_point_sample_vs_reference_golden():
    g = load_golden("point_sample.npz")
    x, c, v = (torch.from_numpy(g[k]).to(DEV) for k in ("input", "coords", "valid"))
    s = A.point_sample(x, c, align_corners=False)
    np.testing.assert_allclose(s.cpu().numpy(), g["sampled"], rtol=1e-5, atol=2e-6)
    m = A.point_sample_masked_mean(x, c, v)
    np.testing.assert_allclose(m.cpu().numpy(), g["pooled"], rtol=1e-5, atol=2e-6)
    grid = A.point_sample(x, c.reshape(3, 8, 5, 2))
    assert grid.shape == (3, 6, 8, 5) and torch.equal(grid.reshape(3, 6, 40), s)


@pytest.mark.parametrize("N,C,H,W,P", [(4, 256, 24, 24, 2304), (1, 3, 1, 1, 7), (2, 17, 5, 9, 1), (3, 8, 6, 4, 0)])
def test_point_sample_vs_oracle(N, C, H, W, P):
    torch.manual_seed(N * 10 + C)
    x = torch.randn(N, C, H, W)
    c = torch.rand(N, P, 2) * 1.3 - 0.15
    v = torch.rand(N, P) > 0.4
    ref = O.point_sample(x, c)
    out = A.point_sample(x.to(DEV), c.to(DEV))
    assert out.shape == (N, C, P)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(A.point_sample_masked_mean(x.to(DEV), c.to(DEV), v.to(DEV)).cpu(), O.masked_mean(ref, v),
                               rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("N,C,H,W,P", [(16, 384, 24, 24, 2304), (3, 100, 5, 7, 300), (2, 9, 80, 80, 500), (2, 200, 24, 24, 64)])
def test_masked_mean_pixel_weight_form(N, C, H, W, P):
    """Round 6: the fused masked mean runs as a pixel-weight product (the points' corner weights scattered once into H W accumulators
    as 2^40-scaled integers, then one dot product per channel) -- the region encoder's shape, an odd map (scalar path: H W % 4 != 0), a
    map too large for the LDS accumulators (the point-walk fallback), few points.  Against the oracle's sample-then-mean; repeated
    launches bit-identical (integer atomics commute); a region without a valid point gives zeros; a NaN at a pixel that no valid
    point touches does not leak through its zero weight."""
    torch.manual_seed(N + C + H)
    x = torch.randn(N, C, H, W)
    c = torch.rand(N, P, 2) * 1.2 - 0.1
    v = torch.rand(N, P) > 0.3
    v[N - 1] = False                                           # the last region: no valid point
    ref = O.masked_mean(O.point_sample(x, c), v)
    out = A.point_sample_masked_mean(x.to(DEV), c.to(DEV), v.to(DEV))
    assert torch.equal(out[N - 1].cpu(), torch.zeros(C)) and torch.isfinite(out).all()
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    for _ in range(3):
        assert torch.equal(A.point_sample_masked_mean(x.to(DEV), c.to(DEV), v.to(DEV)), out)
    # every valid point of region 0 in the lower right quadrant; a NaN in the upper left corner pixel of every plane of region 0
    c2, v2, x2 = c.clone(), v.clone(), x.clone()
    c2[0] = 0.55 + 0.4 * torch.rand(P, 2)
    v2[0] = True
    x2[0, :, 0, 0] = float("nan")
    out2 = A.point_sample_masked_mean(x2.to(DEV), c2.to(DEV), v2.to(DEV))
    assert torch.isfinite(out2[0]).all()
    torch.testing.assert_close(out2[0].cpu(), O.masked_mean(O.point_sample(x, c2), v2)[0], rtol=1e-4, atol=1e-5)


def test_nonfinite_coordinates_and_data():
    x = torch.randn(1, 4, 5, 5)
    c = torch.tensor([[[float("nan"), 0.5], [float("inf"), 0.5], [0.5, -float("inf")], [0.5, 0.5], [5.0, 5.0]]])
    out = A.point_sample(x.to(DEV), c.to(DEV)).cpu()
    assert torch.equal(out[0, :, [0, 1, 2, 4]], torch.zeros(4, 4))          # rejected / far away: zeros, no fault
    torch.testing.assert_close(out[0, :, 3], x[0, :, 2, 2], rtol=1e-6, atol=1e-6)   # centre of the middle pixel
    xn = x.clone()
    xn[0, :, 0, :] = float("nan")                                            # a NaN row that the sample at (2,2) never touches
    assert torch.isfinite(A.point_sample(xn.to(DEV), c[:, 3:4].to(DEV))).all()


def test_errors_and_module():
    x = torch.randn(2, 4, 6, 6)
    with pytest.raises(RuntimeError, match="CUDA"):
        A.point_sample(x, torch.rand(2, 3, 2))
    with pytest.raises(NotImplementedError):
        A.point_sample(x.to(DEV), torch.rand(2, 3, 2, device=DEV), align_corners=True)
    torch.manual_seed(0)
    enc = A.RegionEncoder(hidden_dim=32, embed_dim=16, out_dim=24, patch_size=14, mask_pool_type="grid_sample").to(DEV).eval()
    assert sorted(k for k in enc.state_dict() if k.startswith("up_dim")) == ["up_dim.bias", "up_dim.weight"]
    images = torch.randn(3, 3, 56, 56, device=DEV)
    masks = torch.zeros(3, 1, 56, 56, device=DEV)
    masks[0, 0, 10:30, 5:40] = 1
    masks[1, 0, 0:56, 0:56] = 1                  # region 2 stays empty
    feats = [torch.randn(3, 16, 4, 4, device=DEV), torch.randn(3, 16, 16, device=DEV)]
    torch.manual_seed(1)
    with torch.no_grad():
        out = enc(images, masks, feats)
    assert out.shape == (3, 24) and torch.isfinite(out).all()
    # an empty region pools to zero -> its embedding is the bias of up_dim
    torch.testing.assert_close(out[2], enc.up_dim.bias, rtol=1e-6, atol=1e-6)
    # same random points, pooling restated with the oracle
    torch.manual_seed(1)
    with torch.no_grad():
        f = enc.mask_embedding(torch.cat([images, masks], 1))
        expect = []
        for lvl in feats:
            if lvl.dim() == 3:
                lvl = lvl.reshape(3, 4, 4, -1).permute(0, 3, 1, 2)
            f = f + lvl
            div = torch.tensor([1, 56, 56], device=DEV)[None,]
            pts = torch.nn.utils.rnn.pad_sequence([A.rand_sample(m, div, enc.num_points) for m in masks], padding_value=-1).permute(1, 0, 2)
            valid = pts.sum(-1) >= 0
            s = O.point_sample(f.cpu(), pts[:, :, -2:].flip(dims=[-1]).cpu())
            expect.append(enc.up_dim(O.masked_mean(s, valid.cpu()).to(DEV)))
        expect = torch.stack(expect).mean(0)
    torch.testing.assert_close(out, expect, rtol=1e-4, atol=1e-4)


def test_rand_sample_matches_the_reference_distribution_rule():
    """Every mask channel present gets the same total probability mass; points are returned sorted and unique."""
    m = torch.zeros(2, 8, 8, device=DEV)
    m[0, :2, :] = 1          # 16 points
    m[1, 4, 4] = 1           # 1 point
    div = torch.tensor([1, 8, 8], device=DEV)[None,]
    torch.manual_seed(0)
    pts = A.rand_sample(m, div, 5)
    assert pts.shape == (5, 3) and len({tuple(p.tolist()) for p in pts}) == 5
    hits = 0
    for seed in range(200):
        torch.manual_seed(seed)
        hits += int((A.rand_sample(m, div, 1)[:, 0] == 1).any())
    assert 70 <= hits <= 130    # the single point of mask 1 carries half of the mass
