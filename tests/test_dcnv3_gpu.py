"""DCNv3 forward on the GPU (libvllm_hip.so through the C ABI) against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dcnv3 as O
from visionllm_amd import dcnv3 as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["dcnv3_kat_seed3.npz", "dcnv3_stride2_dil2.npz", "dcnv3_k5x3_odd_channels.npz"]


def _args(g):
    kh, kw, sh, sw, ph, pw, dh, dw, M, Dc = [int(v) for v in g["params"]]
    return (kh, kw, sh, sw, ph, pw, dh, dw, M, Dc, float(g["offset_scale"]))


@pytest.mark.parametrize("name", CASES)
def test_forward_vs_reference_golden(name):
    g = load_golden(name)
    a = _args(g)
    t = lambda k, dt: torch.from_numpy(g[k]).to(dt).to(DEV)
    o64 = A.dcnv3_forward(t("input", torch.float64), t("offset", torch.float64), t("mask", torch.float64), *a, 2)
    np.testing.assert_allclose(o64.cpu().numpy(), g["out_f64"], rtol=1e-5, atol=1e-8)    # test.py:50 (double)
    o32 = A.DCNv3Function.apply(t("input", torch.float32), t("offset", torch.float32), t("mask", torch.float32), *a, 2)
    np.testing.assert_allclose(o32.cpu().numpy(), g["out_f32"], rtol=1e-2, atol=1e-3)    # test.py:77 (float)
    # fp64 kernel against the fp64 C oracle: same arithmetic, only contraction may differ
    ref = O.forward(g["input"].astype(np.float64), g["offset"].astype(np.float64), g["mask"].astype(np.float64), *a)
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("N,H,W,G,C,k,s,p,d,scale", [
    (2, 56, 56, 4, 16, 3, 1, 1, 1, 1.0),     # InternImage stage-1-like (vector path: C % 4 == 0)
    (1, 30, 41, 6, 20, 3, 2, 1, 1, 2.0),     # strided, ragged map
    (2, 17, 13, 3, 7, 5, 1, 2, 1, 0.7),      # scalar path (C % 4 != 0), 5x5
    (1, 9, 9, 2, 8, 3, 1, 2, 2, 1.5),        # dilation 2
    (3, 4, 5, 1, 4, 1, 1, 0, 1, 1.0),        # 1x1 kernel
])
def test_forward_f32_vs_oracle(N, H, W, G, C, k, s, p, d, scale):
    rng = np.random.default_rng(N * 100 + H)
    Ho, Wo = O.out_size(H, W, k, k, s, s, p, p, d, d)
    inp = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
    off = (rng.standard_normal((N, Ho, Wo, G * k * k * 2)) * 2.5).astype(np.float32)   # many samples leave the map
    msk = rng.random((N, Ho, Wo, G * k * k)).astype(np.float32)
    out = A.dcnv3_forward(torch.from_numpy(inp).to(DEV), torch.from_numpy(off).to(DEV), torch.from_numpy(msk).to(DEV),
                          k, k, s, s, p, p, d, d, G, C, scale)
    ref = O.forward(inp, off, msk, k, k, s, s, p, p, d, d, G, C, scale)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    # the reference's CPU path (grid_sample twin) agrees too (pad_h == pad_w)
    tw = O.core_pytorch_twin(torch.from_numpy(inp), torch.from_numpy(off), torch.from_numpy(msk), k, k, s, s, p, p, d, d, G, C,
                             scale).numpy()
    np.testing.assert_allclose(out.cpu().numpy(), tw, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("N,H,W,G,C,k,s,p,d,scale,sigma", [
    (2, 56, 56, 4, 16, 3, 1, 1, 1, 1.0, 1.0),      # InternImage stage-1-like, group channels 16
    (2, 42, 42, 5, 32, 3, 1, 1, 1, 1.0, 1.0),      # group channels 32 (bench shape family)
    (1, 30, 41, 3, 32, 3, 2, 1, 1, 2.0, 1.5),      # strided, ragged map (partial tiles)
    (1, 19, 21, 2, 16, 3, 1, 2, 2, 1.5, 1.0),      # dilation 2
    (2, 11, 9, 2, 32, 1, 1, 0, 1, 1.0, 0.7),       # 1x1 kernel (one point per pixel)
    (1, 20, 20, 2, 32, 2, 1, 0, 1, 1.0, 1.0),      # 2x2 kernel
    (1, 40, 40, 2, 32, 3, 1, 1, 1, 1.0, 9.0),      # wide offsets: windows exceed the LDS budget -> global-memory tiles
    (1, 40, 40, 2, 16, 3, 1, 1, 1, 1.0, 30.0),     # mostly rejected points
    (2, 168, 168, 20, 32, 3, 1, 1, 1, 1.0, 1.0),   # bench shape: every block pipelines ~70 tiles
    (1, 150, 90, 10, 16, 3, 1, 1, 1, 1.0, 2.0),    # several tiles per block, group channels 16, a mix of hot and cold tiles
])
def test_tiled_kernel_vs_oracle_and_gather_kernel(N, H, W, G, C, k, s, p, d, scale, sigma):
    """The LDS-tiled kernels (dcnv3_pipe.hip / dcnv3_tiled.hip; fp32, group channels 16 / 32, <= 9 points) against the oracle and
    against the gather kernel (option dcnv3_tiled = 0); two runs are bit-identical (race screen)."""
    from visionllm_amd import _lib
    rng = np.random.default_rng(H * 7 + C)
    Ho, Wo = O.out_size(H, W, k, k, s, s, p, p, d, d)
    inp = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
    off = (rng.standard_normal((N, Ho, Wo, G * k * k * 2)) * sigma).astype(np.float32)
    off.reshape(-1)[3::101] = np.nan
    off.reshape(-1)[5::103] = np.inf
    msk = rng.random((N, Ho, Wo, G * k * k)).astype(np.float32)
    a = (torch.from_numpy(inp).to(DEV), torch.from_numpy(off).to(DEV), torch.from_numpy(msk).to(DEV), k, k, s, s, p, p, d, d, G, C,
         scale)
    old = _lib.set_option("dcnv3_tiled", 0)
    try:
        plain = A.dcnv3_forward(*a)
        res = {}
        for mode in (1, 3):   # 1: pipelined kernel (dcnv3_pipe.hip, default); 3: two-blocks-per-CU kernel (dcnv3_tiled.hip)
            _lib.set_option("dcnv3_tiled", mode)
            res[mode] = (A.dcnv3_forward(*a), A.dcnv3_forward(*a))
    finally:
        _lib.set_option("dcnv3_tiled", old)
    ref = O.forward(inp, off, msk, k, k, s, s, p, p, d, d, G, C, scale)
    assert np.isfinite(ref).all()
    for mode, (out, again) in res.items():
        assert torch.equal(out, again), f"race: two runs of kernel variant {mode} differ"
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(out, plain, rtol=1e-5, atol=1e-5)


def test_nonfinite_data_and_locations_do_not_leak():
    N, H, W, G, C, k = 1, 6, 6, 2, 4, 3
    inp = torch.randn(N, H, W, G * C)
    off = torch.zeros(N, H, W, G * k * k * 2)
    off[0, 0, 0, :4] = torch.tensor([float("nan"), 0.0, float("inf"), -float("inf")])   # rejected points
    msk = torch.full((N, H, W, G * k * k), 1.0 / 9)
    ref = O.forward(inp.numpy(), off.numpy(), msk.numpy(), k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0)
    assert np.isfinite(ref).all()
    out = A.dcnv3_forward(inp.to(DEV), off.to(DEV), msk.to(DEV), k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    # NaN pixels that no accepted corner touches: the border ring is only reached through clamped addresses
    inp2 = inp.clone()
    off2 = torch.full_like(off, 100.0)                       # every sample far outside -> nothing contributes
    inp2[:] = float("nan")
    out2 = A.dcnv3_forward(inp2.to(DEV), off2.to(DEV), msk.to(DEV), k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0)
    assert torch.equal(out2, torch.zeros_like(out2))


def test_errors_and_module():
    x = torch.randn(2, 8, 8, 32)
    with pytest.raises(RuntimeError, match="Not implement on cpu"):
        A.dcnv3_forward(x, torch.zeros(2, 8, 8, 72), torch.zeros(2, 8, 8, 36), 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 1.0)
    xg = x.to(DEV)
    with pytest.raises(RuntimeError, match="wont match"):
        A.dcnv3_forward(xg, torch.zeros(2, 8, 8, 72, device=DEV), torch.zeros(2, 8, 8, 36, device=DEV), 3, 3, 1, 1, 1, 1, 1, 1, 4,
                        7, 1.0)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        A.dcnv3_forward(torch.randn(3, 8, 8, 32, device=DEV), torch.zeros(3, 8, 8, 72, device=DEV),
                        torch.zeros(3, 8, 8, 36, device=DEV), 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 1.0, 2)
    with pytest.raises(ValueError, match="divisible by group"):
        A.DCNv3(channels=30, group=4)
    torch.manual_seed(0)
    mod = A.DCNv3(channels=32, group=4, offset_scale=1.5, center_feature_scale=True).to(DEV).eval()
    with torch.no_grad():
        mod.offset.weight.normal_(0, 0.3)
        mod.mask.weight.normal_(0, 0.3)
        mod.center_feature_scale_proj_weight.normal_(0, 0.3)
        y = mod(xg)
        # the same module with the sampling core replaced by the oracle
        x1 = mod.dw_conv(xg.permute(0, 3, 1, 2))
        off = mod.offset(x1)
        msk = torch.softmax(mod.mask(x1).reshape(2, 8, 8, 4, -1), -1).reshape(2, 8, 8, -1)
        xp = mod.input_proj(xg)
        core = torch.from_numpy(O.forward(xp.cpu().numpy(), off.cpu().numpy(), msk.cpu().numpy(), 3, 3, 1, 1, 1, 1, 1, 1, 4, 8,
                                          1.5)).to(DEV)
        cfs = torch.sigmoid(torch.nn.functional.linear(x1, mod.center_feature_scale_proj_weight,
                                                       mod.center_feature_scale_proj_bias))
        cfs = cfs[..., None].repeat(1, 1, 1, 1, 8).flatten(-2)
        expect = mod.output_proj(core * (1 - cfs) + xp * cfs)
    torch.testing.assert_close(y, expect, rtol=1e-4, atol=1e-4)
    assert sorted(k for k in mod.state_dict() if "dw_conv" not in k) == sorted([
        "center_feature_scale_proj_bias", "center_feature_scale_proj_weight", "input_proj.bias", "input_proj.weight",
        "mask.bias", "mask.weight", "offset.bias", "offset.weight", "output_proj.bias", "output_proj.weight"])


# ---------------------------------------------------------------------------------------------------------
# round 4: DCNv3 backward (dcnv3_col2im; DCNv3Function.backward, functions/dcnv3_func.py:51-59)
BWD_CASES = CASES + ["dcnv3_bwd_c1.npz", "dcnv3_bwd_c30.npz", "dcnv3_bwd_c71.npz"]


@pytest.mark.parametrize("name", BWD_CASES)
def test_backward_vs_reference_autograd_golden(name):
    """Gradients made by autograd through the reference's dcnv3_core_pytorch (the reference's own backward test, ops_dcnv3/test.py:94-235,
    its thresholds rtol 1e-2 / atol 1e-3) -- through DCNv3Function (autograd) in fp64, and the raw entry point in fp32."""
    g = load_golden(name)
    a = _args(g)
    t = lambda k, dt: torch.from_numpy(g[k]).to(dt).to(DEV)
    inp, off, msk = (t(k, torch.float64).requires_grad_(True) for k in ("input", "offset", "mask"))
    out = A.DCNv3Function.apply(inp, off, msk, *a, 2)
    out.backward(t("grad_out", torch.float64))
    for got, key in ((inp.grad, "grad_input_f64"), (off.grad, "grad_offset_f64"), (msk.grad, "grad_mask_f64")):
        ref = g[key]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-2, atol=1e-3, err_msg=key)                   # the reference's bar
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * np.abs(ref).max() + 1e-12, err_msg=key)   # (fp32 reference points in the twin)
    # fp64 kernel against the fp64 C oracle: the same arithmetic per (point, channel); sums in another order
    d = lambda k: g[k].astype(np.float64)
    oi, oo, om = O.backward(d("input"), d("offset"), d("mask"), d("grad_out"), *a)
    for got, ref, key in ((inp.grad, oi, "grad_input"), (off.grad, oo, "grad_offset"), (msk.grad, om, "grad_mask")):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-10, atol=1e-12 * max(np.abs(ref).max(), 1e-30) + 1e-18, err_msg=key)
    gi, go, gm = A.dcnv3_backward(t("input", torch.float32), t("offset", torch.float32), t("mask", torch.float32), *a,
                                  t("grad_out", torch.float32), 2)
    for got, key in ((gi, "grad_input_f64"), (go, "grad_offset_f64"), (gm, "grad_mask_f64")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-2, atol=1e-3, err_msg=key)                 # test.py:203-235 (float)
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-3, atol=2e-5 * np.abs(g[key]).max(), err_msg=key)


@pytest.mark.parametrize("N,H,W,G,C,k,s,p,d,scale", [
    (2, 56, 56, 4, 16, 3, 1, 1, 1, 1.0),     # InternImage stage-1-like: group channels 16 (4 lanes per group)
    (2, 42, 42, 5, 32, 3, 1, 1, 1, 1.0),     # group channels 32 (8 lanes)
    (1, 20, 24, 3, 64, 3, 1, 1, 1, 1.0),     # 64 (16 lanes): the widest vectorised group
    (1, 30, 41, 6, 20, 3, 2, 1, 1, 2.0),     # 20 channels: multiple of 4 but 5 lanes per group -> the generic kernel; strided, ragged map
    (2, 17, 13, 3, 7, 5, 1, 2, 1, 0.7),      # odd channel count, 5x5
    (1, 9, 9, 2, 8, 3, 1, 2, 2, 1.5),        # dilation 2, 2 lanes per group
    (3, 4, 5, 1, 4, 1, 1, 0, 1, 1.0),        # 1x1 kernel, one lane per group
    (2, 8, 8, 2, 1025, 3, 1, 1, 1, 2.0),     # the reference's gradcheck list ends with 1025 channels (test.py:257)
    # group channels 32: the windowed kernel (msda_bwd_mfma.hip with the DCN flag: S^T x grad_out on the fp32 MFMA)
    (1, 37, 29, 3, 32, 5, 1, 2, 1, 0.7),     # 5 x 5 = 25 points = 7 pseudo-levels of 4, ragged map
    (2, 33, 50, 2, 32, 3, 2, 1, 1, 2.0),     # stride 2: the query grid (17 x 25) is not the input map
    (1, 19, 23, 4, 32, 3, 1, 2, 2, 1.5),     # dilation 2, pad 2
    (2, 9, 11, 2, 32, 1, 1, 0, 1, 1.0),      # 1 x 1 kernel: one point, three rejected slots
    (1, 64, 64, 2, 32, 3, 1, 1, 1, 9.0),     # offset scale 9: windows beyond the staged (512) and the windowed (1024) limits
])
def test_backward_f32_vs_oracle(N, H, W, G, C, k, s, p, d, scale):
    """fp32 kernels against the fp32 C oracle: grad_offset / grad_mask per element (same terms, other summation order over the
    group's channels), grad_input per element within the fp32 rounding of the sum of |terms| (the scatter order is the hardware's);
    NaN / inf offsets are rejected points with zero gradients; two runs agree to the same bound."""
    rng = np.random.default_rng(N * 100 + H + C)
    Ho, Wo = O.out_size(H, W, k, k, s, s, p, p, d, d)
    inp = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
    off = (rng.standard_normal((N, Ho, Wo, G * k * k * 2)) * 2.5).astype(np.float32)   # many samples leave the map
    off.reshape(-1)[5::97] = np.nan
    off.reshape(-1)[11::131] = np.inf
    msk = rng.random((N, Ho, Wo, G * k * k)).astype(np.float32)
    go = rng.standard_normal((N, Ho, Wo, G * C)).astype(np.float32)
    tt = lambda a: torch.from_numpy(a).to(DEV)
    gi, gof, gm = A.dcnv3_backward(tt(inp), tt(off), tt(msk), k, k, s, s, p, p, d, d, G, C, scale, tt(go))
    gi2, gof2, gm2 = A.dcnv3_backward(tt(inp), tt(off), tt(msk), k, k, s, s, p, p, d, d, G, C, scale, tt(go))
    ri, ro, rm = O.backward(inp, off, msk, go, k, k, s, s, p, p, d, d, G, C, scale)
    # magnitude of the terms that enter an element of grad_input: the same scatter with absolute values
    mi, _, _ = O.backward(np.abs(inp), off, np.abs(msk), np.abs(go), k, k, s, s, p, p, d, d, G, C, scale)
    assert torch.isfinite(gi).all() and torch.isfinite(gof).all() and torch.isfinite(gm).all()
    assert torch.equal(gof, gof2) and torch.equal(gm, gm2)                     # fixed reduction trees
    assert (np.abs(gi.cpu().numpy() - ri) <= 2.0 ** -18 * mi + 1e-7).all()
    assert (np.abs(gi2.cpu().numpy() - gi.cpu().numpy()) <= 2.0 ** -18 * mi + 1e-7).all()
    sc_o, sc_m = np.abs(ro).max() + 1e-12, np.abs(rm).max() + 1e-12
    np.testing.assert_allclose(gof.cpu().numpy(), ro, rtol=2e-4, atol=3e-6 * sc_o)
    np.testing.assert_allclose(gm.cpu().numpy(), rm, rtol=2e-4, atol=3e-6 * sc_m)


def test_backward_windowed_kernel_agrees_with_gather_kernel():
    """Option "dcnv3_bwd_tiled" 1 (default: the windowed MFMA kernel for group channels 32) against 0 (one thread group per (pixel,
    group), global atomics per (point, corner, channel)): the same gradients to fp32 rounding, at an InternImage-like stage."""
    from visionllm_amd import _lib
    torch.manual_seed(3)
    N, H, W, G, C, k = 2, 84, 84, 4, 32, 3
    x = torch.randn(N, H, W, G * C, device=DEV)
    off = torch.randn(N, H, W, G * k * k * 2, device=DEV) * 1.5
    m = torch.softmax(torch.randn(N, H, W, G, k * k, device=DEV), -1).reshape(N, H, W, -1)
    go = torch.randn(N, H, W, G * C, device=DEV)
    res = {}
    old = _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", 0)
    try:
        for v in (0, 1):
            _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", v)
            res[v] = A.dcnv3_backward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, C, 1.0, go)
    finally:
        _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", old)
    for a, b, what in zip(res[0], res[1], ("grad_input", "grad_offset", "grad_mask")):
        sc = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * sc, (what, float((a - b).abs().max()), sc)


@pytest.mark.parametrize("scale", [1.0, 1.7])
def test_locations_on_and_next_to_integers_land_in_the_same_cell_in_every_kernel(scale):
    """a location within an ulp of an integer must fall into the SAME cell in the gather backward, the windowed matrix-core
    backward and the oracle (the reference's operation order: every product rounded on its own, `dcn_loc`): grad_offset is
    discontinuous there, so one fused multiply-add in one kernel shows as an O(1) difference.  Offsets are constructed so that
    p0 + (i d + offset) scale is an integer, or its fp32 neighbour above / below, for every point."""
    from visionllm_amd import _lib
    N, H, W, G, C, k = 1, 24, 40, 2, 32, 3
    rng = np.random.default_rng(5)
    inp = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
    msk = rng.random((N, H, W, G * k * k)).astype(np.float32)
    go = rng.standard_normal((N, H, W, G * C)).astype(np.float32)
    # target: an integer 0 .. 2 cells away from the undeformed point; offset = target / scale (then nudged by -1 / 0 / +1 ulp)
    tgt = rng.integers(-2, 3, size=(N, H, W, G * k * k * 2)).astype(np.float32)
    off = (tgt / np.float32(scale)).astype(np.float32)
    nudge = rng.integers(-1, 2, size=off.shape)
    off = np.where(nudge > 0, np.nextafter(off, np.float32(np.inf)), np.where(nudge < 0, np.nextafter(off, np.float32(-np.inf)), off)).astype(np.float32)
    tt = lambda a: torch.from_numpy(a).to(DEV)
    res = {}
    old = _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", 0)
    try:
        for v in (0, 1):
            _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", v)
            res[v] = [g.cpu().numpy() for g in A.dcnv3_backward(tt(inp), tt(off), tt(msk), k, k, 1, 1, 1, 1, 1, 1, G, C, scale, tt(go))]
    finally:
        _lib.lib().vllm_set_option(b"dcnv3_bwd_tiled", old)
    ri, ro, rm = O.backward(inp, off, msk, go, k, k, 1, 1, 1, 1, 1, 1, G, C, scale)
    sc_o, sc_m = np.abs(ro).max(), np.abs(rm).max()
    for v in (0, 1):
        np.testing.assert_allclose(res[v][1], ro, rtol=2e-4, atol=1e-5 * sc_o, err_msg=f"grad_offset, dcnv3_bwd_tiled = {v}")
        np.testing.assert_allclose(res[v][2], rm, rtol=2e-4, atol=1e-5 * sc_m, err_msg=f"grad_mask, dcnv3_bwd_tiled = {v}")
    # and the forward kernels (continuous in the location, but a wrong cell at a map border drops or adds a corner)
    ref = O.forward(inp, off, msk, k, k, 1, 1, 1, 1, 1, 1, G, C, scale)
    old = _lib.set_option("dcnv3_tiled", 0)
    try:
        for mode in (0, 1, 3):
            _lib.set_option("dcnv3_tiled", mode)
            out = A.dcnv3_forward(tt(inp), tt(off), tt(msk), k, k, 1, 1, 1, 1, 1, 1, G, C, scale)
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5, err_msg=f"forward, dcnv3_tiled = {mode}")
    finally:
        _lib.set_option("dcnv3_tiled", old)


def test_backward_argument_checks():
    x = torch.zeros(1, 4, 4, 8, device=DEV)
    off = torch.zeros(1, 4, 4, 2 * 9 * 2, device=DEV)
    msk = torch.zeros(1, 4, 4, 2 * 9, device=DEV)
    go = torch.zeros(1, 4, 4, 8, device=DEV)
    with pytest.raises(RuntimeError, match="Not implement on cpu"):
        A.dcnv3_backward(x.cpu(), off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 2, 4, 1.0, go)
    with pytest.raises(RuntimeError, match="wont match"):
        A.dcnv3_backward(x, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 3, 4, 1.0, go)
    with pytest.raises(RuntimeError, match="geometry"):
        A.dcnv3_backward(x, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 2, 4, 1.0, go[:, :3].contiguous())
    gi, gof, gm = A.dcnv3_backward(x, off, msk, 3, 3, 1, 1, 1, 1, 1, 1, 2, 4, 1.0, go)
    assert gi.shape == x.shape and gof.shape == off.shape and gm.shape == msk.shape and float(gi.abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["c16", "c32", "c5"])
def test_half_precision_forward_and_backward(tag):
    """Round 5: float16 operands (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF with opmath_t = float,
    ops_dcnv3/src/cuda/dcnv3_cuda.cu:69, 147).  Three checks: (1) against the fixture made from the reference's twin on the widened
    operands (tests/golden/dcnv3_half.npz, one half ulp: both sides round an fp32 result); (2) BIT-EXACT against this library's own
    fp32 kernels on the widened operands rounded to half -- the half entry points add nothing but the conversions; (3) autograd through
    DCNv3Function in half."""
    g = load_golden("dcnv3_half.npz")
    kh, kw, sh, sw, ph, pw, dh, dw, M, Dc = [int(v) for v in g[f"{tag}.params"]]
    a = (kh, kw, sh, sw, ph, pw, dh, dw, M, Dc, float(g[f"{tag}.offset_scale"]))
    h = lambda k: torch.from_numpy(g[f"{tag}.{k}"]).to(DEV)
    inp, off, msk, go = h("input"), h("offset"), h("mask"), h("grad_out")
    assert inp.dtype == torch.float16
    out = A.dcnv3_forward(inp, off, msk, *a)
    assert out.dtype == torch.float16
    np.testing.assert_allclose(out.float().cpu().numpy(), g[f"{tag}.out"].astype(np.float32), rtol=1e-3, atol=1e-3)
    from visionllm_amd import _lib
    old = _lib.set_option("dcnv3_tiled", 0)     # the gather kernel: the half kernel is its instantiation
    try:
        out32 = A.dcnv3_forward(inp.float(), off.float(), msk.float(), *a)
    finally:
        _lib.set_option("dcnv3_tiled", old)
    assert torch.equal(out, out32.half())
    gi, gof, gm = A.dcnv3_backward(inp, off, msk, *a, go)
    for ours, key in ((gi, "grad_input"), (gof, "grad_offset"), (gm, "grad_mask")):
        ref = g[f"{tag}.{key}"].astype(np.float32)
        assert ours.dtype == torch.float16
        np.testing.assert_allclose(ours.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * max(np.abs(ref).max(), 1e-3), err_msg=key)
    gi32, gof32, gm32 = A.dcnv3_backward(inp.float(), off.float(), msk.float(), *a, go.float())
    assert torch.equal(gof, gof32.half()) and torch.equal(gm, gm32.half())
    torch.testing.assert_close(gi.float(), gi32, rtol=1e-3, atol=1e-3 * float(gi32.abs().max()))    # (atomics: the fp32 sums differ in the last bits run to run)
    x = inp.clone().requires_grad_(True)
    o2 = A.DCNv3Function.apply(x, off, msk, *a, 256)
    o2.backward(go)
    assert torch.equal(o2.detach(), out) and x.grad.dtype == torch.float16
    torch.testing.assert_close(x.grad.float(), gi.float(), rtol=1e-3, atol=1e-3 * float(gi32.abs().max()))
