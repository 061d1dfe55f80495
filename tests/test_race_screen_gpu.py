"""Race screens in the driver-run suite (round-4 review, item 2).

The MSDA forward the library chooses on its own (generation 9: inline-asm LDS reads with counted waits, two teams sharing one
arena), the streaming K = 256 GEMMs of the deformable-attention layer (same read / wait pattern) and the persistent GEMM with a
half-height last round are launched MANY times on fresh output tensors; every result must equal the first bit for bit.  The round-4
generation-9 race showed on 1 - 8 launches of 150 - 300 and only on location sets that mix staged, late and global-memory levels: a
two-run identity check passes that 98 % of the time, a hundred launches do not.  The reference's call sites swallow kernel errors
(modeling_ov_grounding_dino_mask_dn.py:767-779), so a flaky kernel would be invisible downstream.

The automatic choice is also compared against the two kernels with plain, compiler-visible reads that serve the same shape (generation 4
and the gather kernel; 2e-6: only the association of the weighted sum differs), so a toolchain change that breaks generation 9's
register assumptions shows as a difference (last item; generation 8 left the library in round 5).  The backward the
library chooses gets the same screen (40 launches per geometry).  Whole file: ~40 s on MI355X.
"""
import math

import numpy as np
import pytest
import torch

from msda_inputs import CFG4_SHAPES, make_inputs
from test_msda_gpu import PYRAMIDS
from visionllm_amd import _lib
from visionllm_amd import ms_deform_attn as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_LAUNCH = 100


def _mixed(shapes, seed):
    """Locations that exposed the round-4 race: a third of the points far away (late + global-memory levels), rejected,
    NaN and inf points in between (tools/gpu_passes/dbg_msda9_race.py)."""
    g = make_inputs(2, 8, 32, shapes, 4, mode="encoder_like", seed=seed)
    rng = np.random.default_rng(7)
    loc = g["loc"].copy()
    flat = loc.reshape(-1, 2)
    flat[1::3] += rng.standard_normal(flat[1::3].shape).astype(np.float32) * 0.15
    flat[3::29] = 1.7
    flat[5::97] = np.nan
    flat[6::101] = np.inf
    g["loc"] = loc
    return g


def _dev(g):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in g.items()}


def _fwd(t):
    return A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)


@pytest.mark.parametrize("name", sorted(PYRAMIDS))
def test_msda_automatic_forward_100_launches_identical(name):
    g = _mixed(PYRAMIDS[name], seed=len(PYRAMIDS[name]))
    old = _lib.set_option("msda_tiled", 1)          # the automatic choice
    try:
        first = _fwd(_dev(g))
        for i in range(N_LAUNCH):
            out = _fwd(_dev(g))                     # fresh tensors every launch: new addresses, cold caches (as the tests do)
            assert torch.equal(out, first), f"{name}: launch {i} differs from the first (max {float((out - first).abs().max()):.3g})"
        _lib.set_option("msda_tiled", 9)            # generation 4 (any geometry): plain compiler-visible LDS reads, same arithmetic per point
        torch.testing.assert_close(first, _fwd(_dev(g)), rtol=2e-6, atol=2e-6)
        _lib.set_option("msda_tiled", 0)            # gather kernel
        torch.testing.assert_close(first, _fwd(_dev(g)), rtol=2e-6, atol=2e-6)
    finally:
        _lib.set_option("msda_tiled", old)


def test_msda_cfg4_batch8_mixed_40_launches_identical():
    g = make_inputs(8, 8, 32, CFG4_SHAPES, 4, mode="encoder_like", seed=11)
    rng = np.random.default_rng(3)
    flat = g["loc"].reshape(-1, 2)
    flat[2::5] += rng.standard_normal(flat[2::5].shape).astype(np.float32) * 0.1
    t = _dev(g)
    old = _lib.set_option("msda_tiled", 1)
    try:
        first = _fwd(t)
        for i in range(40):
            out = _fwd(t)
            assert torch.equal(out, first), f"launch {i} differs"
    finally:
        _lib.set_option("msda_tiled", old)


@pytest.mark.parametrize("name", sorted(PYRAMIDS))
def test_msda_backward_40_launches_identical(name):
    """The backward the library chooses (round 5: the matrix-core kernel was restructured -- per-level barriers removed, S^T zeroed by its
    readers, the window copied by LDS-DMA behind the scatter): grad_sampling_loc / grad_attn_weight have one owner per element and
    must repeat bit for bit; grad_value is accumulated with atomics (order varies: 2^-18 of the magnitude sum, the bound of
    test_msda_gpu's backward tests); all three against the kernel without windows (msda_tiled = 0)."""
    g = _mixed(PYRAMIDS[name], seed=len(PYRAMIDS[name]) + 1)
    t = _dev(g)
    torch.manual_seed(3)
    go = torch.randn(t["loc"].shape[0], t["loc"].shape[1], t["value"].shape[2] * t["value"].shape[3], device=DEV)
    bwd = lambda: A.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], go, 64)
    old = _lib.set_option("msda_tiled", 1)
    try:
        gv0, gl0, gw0 = bwd()
        assert bool(torch.isfinite(gv0).all()) and bool(torch.isfinite(gl0).all()) and bool(torch.isfinite(gw0).all())
        tol = 2.0 ** -18 * float(gv0.abs().max()) * 16 + 1e-7
        for i in range(40):
            gv, gl, gw = bwd()
            assert torch.equal(gl, gl0) and torch.equal(gw, gw0), f"{name}: launch {i}: a per-point gradient differs from the first launch"
            assert float((gv - gv0).abs().max()) <= tol, f"{name}: launch {i}: grad_value differs by {float((gv - gv0).abs().max()):.3g}"
        _lib.set_option("msda_tiled", 0)
        gv, gl, gw = bwd()
        torch.testing.assert_close(gw, gw0, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(gl, gl0, rtol=2e-3, atol=2e-3)
        assert float((gv - gv0).abs().max()) <= 64 * tol
    finally:
        _lib.set_option("msda_tiled", old)


@pytest.mark.parametrize("name,M,N,K,epi", [("qkv_40_tiles_half_tail", 23080, 3072, 1024, 0), ("qkv_32_tiles_gelu", 18464, 3072, 1024, 1),
                                            ("two_k_tiles_ragged", 2 * 256 * 4 + 130, 2048, 128, 0)])
def test_gemm_persistent_and_half_tail_100_launches_identical(name, M, N, K, epi):
    L, st, P = _lib.lib(), _lib.current_stream(), _lib.ptr
    torch.manual_seed(0)
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    first = None
    for i in range(N_LAUNCH):
        y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, None, 0, 0, st))
        if first is None:
            first = y
            assert bool(torch.isfinite(first.float()).all())
        else:
            assert torch.equal(first, y), f"{name}: launch {i} differs"


@pytest.mark.parametrize("name,M,epi,masked", [("bias_bf16", 70001, 0, False), ("f32_out", 70001, 5, False), ("f32_out_row_mask", 70000, 5, True)])
def test_gemm_skinny_100_launches_identical(name, M, epi, masked):
    L, st, P = _lib.lib(), _lib.current_stream(), _lib.ptr
    K = N = 256
    torch.manual_seed(1)
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.06).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    mask = (torch.rand(M, device=DEV) < 0.2).to(torch.uint8) if masked else None
    first = None
    for i in range(N_LAUNCH):
        y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32 if epi == 5 else torch.bfloat16)
        _lib.check(L.vllm_gemm_bf16(P(x), P(w), P(b), P(y), M, N, K, K, K, N, epi, None, P(mask) if masked else None, 0, 0, st))
        if first is None:
            first = y
        else:
            assert torch.equal(first, y), f"{name}: launch {i} differs"


def test_fused_layer_60_launches_identical():
    B, C, Mh, Lv, Pp = 2, 256, 8, 4, 4
    S = sum(h * w for h, w in CFG4_SHAPES)
    torch.manual_seed(2)
    mod = A.MSDeformAttn(C, Lv, Mh, Pp).to(DEV).to(torch.bfloat16).eval()
    ss = torch.tensor(CFG4_SHAPES, device=DEV)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    src = torch.randn(B, S, C, device=DEV).bfloat16()
    q = torch.randn(B, S, C, device=DEV).bfloat16()
    ref = torch.rand(B, S, Lv, 2, device=DEV)
    mask = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    mask[-1, -S // 7:] = True
    first = None
    with torch.no_grad():
        for i in range(60):
            o = mod(q, ref, src, ss, lsi, mask)
            if first is None:
                first = o.clone()
            else:
                assert torch.equal(first, o), f"layer launch {i} differs"
