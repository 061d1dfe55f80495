"""Region-encoder point sampling oracle against the golden vectors produced by the reference's own point_sample
(oracle/gen_golden.py: gen_point_sample)."""
import numpy as np

from conftest import load_golden
from oracle import region as R


def test_point_sample_oracle_vs_reference():
    g = load_golden("point_sample.npz")
    s = R.point_sample(g["input"], g["coords"])
    np.testing.assert_allclose(s.numpy(), g["sampled"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(R.masked_mean(s, g["valid"]).numpy(), g["pooled"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(R.point_sample_numpy(g["input"], g["coords"]), g["sampled"], rtol=1e-5, atol=1e-6)
    assert (g["pooled"][2] == 0).all()      # a region without points pools to zero (nan_to_num)
