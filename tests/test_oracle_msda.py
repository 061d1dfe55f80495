"""The C oracle is pinned against the reference's own known-answer inputs/outputs (CPU, no GPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import msda as O

CASES = ["msda_kat_seed3.npz", "msda_stress_small.npz", "msda_stress_d32.npz", "msda_odd_channels.npz"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_forward_f64_matches_reference(name):
    g = load_golden(name)
    out = O.forward(g["value"].astype(np.float64), g["shapes"], g["lsi"], g["loc"].astype(np.float64),
                    g["attw"].astype(np.float64))
    ref = g["out_f64"]
    # reference double test: max_abs_err < 1e-18 / rel < 1e-15 on values ~1e-3 (test_ms_deformable_attn.py:99-102);
    # grid_sample and the kernel order their flops differently, so we allow a few ulp on O(1) data.
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", CASES)
def test_oracle_forward_f32_matches_reference(name):
    g = load_golden(name)
    out = O.forward(g["value"], g["shapes"], g["lsi"], g["loc"], g["attw"])
    # reference float test: allclose(rtol=1e-2, atol=1e-3), max_abs<1e-9 & rel<1e-6 on ~1e-3 data (:129-134)
    scale = np.abs(g["out_f64"]).max()
    assert np.abs(out - g["out_f64"]).max() <= 2e-6 * max(scale, 1e-3)


def test_kat_seed3_literal_values():
    """The literal numbers quoted in SURVEY.md section 8c (evaluated from the reference test recipe)."""
    g = load_golden("msda_kat_seed3.npz")
    out = O.forward(g["value"].astype(np.float64), g["shapes"], g["lsi"], g["loc"].astype(np.float64),
                    g["attw"].astype(np.float64))
    lit = np.array([[0.001899378416, 0.004602827533, 0.004671175247, 0.004384399819],
                    [0.003795097174, 0.002512764199, 0.001844426151, 0.003634679248]])
    np.testing.assert_allclose(out[0], lit, rtol=0, atol=1e-12)


def test_grid_sample_twin_matches_golden():
    for name in CASES:
        g = load_golden(name)
        out = O.grid_sample_twin(torch.from_numpy(g["value"]), g["shapes"].tolist(), torch.from_numpy(g["loc"]),
                                 torch.from_numpy(g["attw"]))
        np.testing.assert_allclose(out.numpy(), g["out_f32"], rtol=1e-5, atol=1e-6)


def test_sample_index_agrees_with_float_floor():
    g = load_golden("msda_stress_d32.npz")
    h, w, mk = O.sample_index(g["shapes"], g["loc"])
    L = g["shapes"].shape[0]
    for l in range(L):
        H, W = g["shapes"][l]
        y = (g["loc"][:, :, :, l, :, 1] * np.float32(H)).astype(np.float32) - np.float32(0.5)
        x = (g["loc"][:, :, :, l, :, 0] * np.float32(W)).astype(np.float32) - np.float32(0.5)
        ok = (y > -1) & (x > -1) & (y < H) & (x < W)
        assert np.array_equal((mk[:, :, :, l] & 1).astype(bool), ok)
        assert np.array_equal(h[:, :, :, l][ok], np.floor(y[ok]).astype(np.int32))
        assert np.array_equal(w[:, :, :, l][ok], np.floor(x[ok]).astype(np.int32))
    assert (mk & 1).sum() > 0 and (mk & 1).sum() < mk.size  # both accepted and rejected points are exercised


def test_oracle_backward_matches_autograd_of_twin():
    g = load_golden("msda_stress_small.npz")
    v = torch.from_numpy(g["value"]).double().requires_grad_(True)
    loc = torch.from_numpy(g["loc"]).double().requires_grad_(True)
    w = torch.from_numpy(g["attw"]).double().requires_grad_(True)
    out = O.grid_sample_twin(v, g["shapes"].tolist(), loc, w)
    torch.manual_seed(0)
    go = torch.randn_like(out)
    out.backward(go)
    gv, gl, gw = O.backward(g["value"].astype(np.float64), g["shapes"], g["lsi"], g["loc"].astype(np.float64),
                            g["attw"].astype(np.float64), go.numpy())
    np.testing.assert_allclose(gv, v.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gw, w.grad.numpy(), rtol=1e-9, atol=1e-12)
    # d/dloc is discontinuous exactly at pixel borders; the fixture forces a few such points -> compare elsewhere
    y = loc.detach().numpy()
    L = g["shapes"].shape[0]
    smooth = np.ones(y.shape[:-1], dtype=bool)
    for l in range(L):
        H, W = g["shapes"][l]
        fy = y[:, :, :, l, :, 1] * H - 0.5
        fx = y[:, :, :, l, :, 0] * W - 0.5
        near = (np.abs(fy - np.round(fy)) < 1e-6) | (np.abs(fx - np.round(fx)) < 1e-6)
        smooth[:, :, :, l] &= ~near
    np.testing.assert_allclose(gl[smooth], loc.grad.numpy()[smooth], rtol=1e-8, atol=1e-10)


def test_empty_query_and_batch():
    shapes = np.array([[2, 3]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    out = O.forward(np.zeros((1, 6, 2, 4), np.float32), shapes, lsi, np.zeros((1, 0, 2, 1, 2, 2), np.float32),
                    np.zeros((1, 0, 2, 1, 2), np.float32))
    assert out.shape == (1, 0, 8)


# ---- the deformable-attention LAYER: oracle.layer_forward vs the reference's own module classes (gen_golden.gen_msda_layer) ----
def _layer_params(g, prefix):
    return {n: (g[f"{prefix}.sd.{n}.weight"], g[f"{prefix}.sd.{n}.bias"])
            for n in ("value_proj", "sampling_offsets", "attention_weights", "output_proj")}


@pytest.mark.parametrize("tag", ["unipose_ref2", "unipose_ref4", "unipose_ref4_norm"])
def test_layer_oracle_vs_reference_unipose_module(tag):
    """MSDeformAttn.forward (unipose/ops/modules/ms_deform_attn.py:83-145): 2-d / 4-d reference points, use_4D_normalizer."""
    g = load_golden("msda_layer.npz")
    M, L, P = int(g["n_heads"]), int(g["n_levels"]), int(g["n_points"])
    out, _, _ = O.layer_forward(g[f"{tag}.query"], g[f"{tag}.ref"], g[f"{tag}.src"], g["shapes"], g["lsi"], g[f"{tag}.mask"],
                                _layer_params(g, tag), M, L, P, use_4d_normalizer=bool(g[f"{tag}.use4d"]))
    # (the reference module runs its operator in fp32 whatever the module dtype -- "for mixed precision",
    # ms_deform_attn.py:131-139 -- so its float64 output carries fp32 rounding of the sampling core)
    np.testing.assert_allclose(out, g[f"{tag}.out_f64"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("tag", ["mmcv_ref2", "mmcv_ref4"])
def test_layer_oracle_vs_reference_mmcv_module(tag):
    """mmcv MultiScaleDeformableAttention.forward (multi_scale_deform_attn.py:262-367): query_pos added to the query,
    (num_query, bs, C) layout, identity residual (dropout is the identity in eval mode)."""
    g = load_golden("msda_layer.npz")
    M, L, P = int(g["n_heads"]), int(g["n_levels"]), int(g["n_points"])
    q = g[f"{tag}.query"].astype(np.float64)
    out, _, _ = O.layer_forward(q + g[f"{tag}.query_pos"], g[f"{tag}.ref"], g[f"{tag}.src"], g["shapes"], g["lsi"],
                                g[f"{tag}.mask"], _layer_params(g, tag), M, L, P)
    np.testing.assert_allclose((out + q).transpose(1, 0, 2), g[f"{tag}.out_f64"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("tag", ["gdino_ref2", "gdino_ref4"])
def test_layer_oracle_vs_reference_grounding_dino_module(tag):
    """GroundingDinoMultiscaleDeformableAttention.forward (...mask_dn.py:706-784): position embeddings, inverted mask."""
    g = load_golden("msda_layer.npz")
    M, L, P = int(g["n_heads"]), int(g["n_levels"]), int(g["n_points"])
    out, _, aw = O.layer_forward(g[f"{tag}.query"].astype(np.float64) + g[f"{tag}.pos"], g[f"{tag}.ref"], g[f"{tag}.src"],
                                 g["shapes"], g["lsi"], g[f"{tag}.mask"], _layer_params(g, tag), M, L, P)
    np.testing.assert_allclose(out, g[f"{tag}.out_f64"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(aw, g[f"{tag}.attw_f32"], rtol=2e-5, atol=2e-6)
