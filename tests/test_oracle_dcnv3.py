"""DCNv3 oracle (oracle/dcnv3_oracle.c + the torch twin) against golden vectors produced by RUNNING the reference's
dcnv3_core_pytorch on its own test inputs (oracle/gen_golden.py: gen_dcnv3; ops_dcnv3/test.py:19-66, seed 3)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dcnv3 as D

CASES = ["dcnv3_kat_seed3.npz", "dcnv3_stride2_dil2.npz", "dcnv3_k5x3_odd_channels.npz"]


def _args(g):
    kh, kw, sh, sw, ph, pw, dh, dw, M, Dc = [int(v) for v in g["params"]]
    return (kh, kw, sh, sw, ph, pw, dh, dw, M, Dc, float(g["offset_scale"]))


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_reference_twin(name):
    g = load_golden(name)
    a = _args(g)
    o64 = D.forward(g["input"].astype(np.float64), g["offset"].astype(np.float64), g["mask"].astype(np.float64), *a)
    # the reference's own tolerance for CUDA kernel vs twin in double (test.py:50: torch.allclose defaults); the twin builds
    # its reference points in float32 (dcnv3_func.py:66-90), so it is not more accurate than that
    np.testing.assert_allclose(o64, g["out_f64"], rtol=1e-5, atol=1e-8)
    o32 = D.forward(g["input"], g["offset"], g["mask"], *a)
    np.testing.assert_allclose(o32, g["out_f32"], rtol=1e-2, atol=1e-3)   # test.py:77 (float)
    np.testing.assert_allclose(o32, g["out_f64"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("name", CASES)
def test_torch_twin_restatement_is_the_reference_twin(name):
    g = load_golden(name)
    a = _args(g)
    t = lambda k: torch.from_numpy(g[k]).double()
    out = D.core_pytorch_twin(t("input"), t("offset"), t("mask"), *a).numpy()
    np.testing.assert_allclose(out, g["out_f64"], rtol=0, atol=1e-15)


def test_reference_known_answer_values():
    """First output values of the reference twin on the seed-3 inputs of ops_dcnv3/test.py (fp64)."""
    g = load_golden("dcnv3_kat_seed3.npz")
    np.testing.assert_allclose(g["out_f64"][0, 0, 0, :3],
                               [0.0003553824683531446, 0.0004284441152049652, 0.00037868138803360803], rtol=1e-12)


BWD_CASES = CASES + ["dcnv3_bwd_c1.npz", "dcnv3_bwd_c30.npz", "dcnv3_bwd_c71.npz"]


@pytest.mark.parametrize("name", BWD_CASES)
def test_c_oracle_backward_matches_reference_autograd(name):
    """Round 4: the C restatement of dcnv3_col2im (dcnv3_im2col_cuda.cuh:86-146, 279-857) against gradients made by autograd through the
    reference's dcnv3_core_pytorch (gen_golden.py::gen_dcnv3; the reference's own backward test, ops_dcnv3/test.py:94-235, compares the
    same two with rtol 1e-2 / atol 1e-3 -- the fp64 restatement is held to 1e-5 relative: the twin builds its reference points in fp32)."""
    g = load_golden(name)
    a = _args(g)
    d = lambda k: g[k].astype(np.float64)
    gi, go, gm = D.backward(d("input"), d("offset"), d("mask"), d("grad_out"), *a)
    for ours, key in ((gi, "grad_input_f64"), (go, "grad_offset_f64"), (gm, "grad_mask_f64")):
        ref = g[key]
        scale = np.abs(ref).max()
        np.testing.assert_allclose(ours, ref, rtol=1e-5, atol=2e-6 * max(scale, 1e-30) + 1e-12, err_msg=key)   # (fp32 reference points in the twin: ~1e-7 * 8 px of location error)
    f = lambda k: g[k].astype(np.float32)
    gi, go, gm = D.backward(f("input"), f("offset"), f("mask"), f("grad_out"), *a)
    for ours, key in ((gi, "grad_input_f64"), (go, "grad_offset_f64"), (gm, "grad_mask_f64")):
        np.testing.assert_allclose(ours, g[key], rtol=1e-2, atol=1e-3, err_msg=key)          # the reference's float thresholds (test.py:203-235)
        np.testing.assert_allclose(ours, g[key], rtol=2e-3, atol=2e-5 * np.abs(g[key]).max(), err_msg=key)


HALF_TAGS = ["c16", "c32", "c5"]


@pytest.mark.parametrize("tag", HALF_TAGS)
def test_c_oracle_on_half_operands_matches_reference_twin(tag):
    """Round 5 (half precision, dcnv3_cuda.cu:69 / :147 dispatch AND_HALF with fp32 arithmetic): the fp32 C restatement on the WIDENED
    half operands, rounded to half, against the fixture made by the reference's twin the same way (gen_golden.py::gen_dcnv3_half).
    Both round an fp32 result once: they may differ by one half ulp where the fp32 values straddle a rounding boundary."""
    g = load_golden("dcnv3_half.npz")
    kh, kw, sh, sw, ph, pw, dh, dw, M, Dc = [int(v) for v in g[f"{tag}.params"]]
    a = (kh, kw, sh, sw, ph, pw, dh, dw, M, Dc, float(g[f"{tag}.offset_scale"]))
    w = lambda k: g[f"{tag}.{k}"].astype(np.float32)
    out = D.forward(w("input"), w("offset"), w("mask"), *a)
    np.testing.assert_allclose(out, g[f"{tag}.out_f32"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.astype(np.float16).astype(np.float32), w("out"), rtol=1e-3, atol=1e-3)
    gi, go, gm = D.backward(w("input"), w("offset"), w("mask"), w("grad_out"), *a)
    for ours, key in ((gi, "grad_input"), (go, "grad_offset"), (gm, "grad_mask")):
        ref = w(key)
        np.testing.assert_allclose(ours.astype(np.float16).astype(np.float32), ref, rtol=2e-3, atol=2e-3 * max(np.abs(ref).max(), 1e-3), err_msg=key)
