"""CPU: the C-ABI library builds, loads and exports every symbol include/vllm_hip.h declares (no compute)."""
import ctypes
import os
import subprocess

import pytest

from visionllm_amd import _lib


def test_header_parses_and_names_are_prefixed():
    protos = _lib.parse_header()
    assert "vllm_msda_forward_f32" in protos and "vllm_last_error" in protos
    assert all(n.startswith("vllm_") for n in protos)


def test_library_exports_every_declared_symbol():
    path = _lib.lib_path()
    if not os.path.exists(path):
        _lib.build()
    raw = ctypes.CDLL(path)
    missing = [n for n in _lib.parse_header() if not hasattr(raw, n)]
    assert not missing, missing
    L = _lib.lib()
    assert L.vllm_abi_version() == 1


def test_no_undeclared_exports():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_path()], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T vllm_" in line}
    assert exported == set(_lib.parse_header()), exported ^ set(_lib.parse_header())


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dirpath, _, files in os.walk(os.path.join(root, "visionllm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                if "import oracle" in src or "from oracle" in src or "oracle/" in src.replace("oracle/ ", ""):
                    bad.append(f)
    assert not bad, bad


def test_ops_fail_loudly_without_gpu_tensors():
    import torch
    from visionllm_amd.ms_deform_attn import ms_deform_attn_forward
    v = torch.zeros(1, 6, 2, 4)
    with pytest.raises(RuntimeError):
        ms_deform_attn_forward(v, torch.tensor([[2, 3]]), torch.tensor([0]), torch.zeros(1, 1, 2, 1, 2, 2),
                               torch.zeros(1, 1, 2, 1, 2), 64)


def test_module_constructor_errors_mirror_reference():
    # mmcv/tests/test_ops/test_ms_deformable_attn.py:25-30
    from visionllm_amd.ms_deform_attn import MSDeformAttn, MultiScaleDeformableAttention
    with pytest.raises(ValueError):
        MultiScaleDeformableAttention(embed_dims=256, num_heads=7)
    with pytest.raises(ValueError):
        MSDeformAttn(d_model=256, n_heads=7)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without the built .so every op raises (the product never routes through the oracle)."""
    import torch
    from visionllm_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "lib_path", lambda: str(tmp_path / "libvllm_hip.so"))
    with pytest.raises(RuntimeError, match="has not been built"):
        L.lib()
    from visionllm_amd.bridge import pixel_shuffle
    with pytest.raises(RuntimeError):
        pixel_shuffle(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16))


def test_level_pixels_is_remembered_per_tensor_object_and_version():
    """ms_deform_attn.level_pixels: the reference's per-call `(shapes[:, 0] * shapes[:, 1]).sum() == Len_in` check, remembered
    per tensor object + autograd version (one host synchronisation per forward pass instead of one per layer)."""
    import torch
    from visionllm_amd import ms_deform_attn as A
    ss = torch.tensor([[3, 4], [2, 2]])
    assert A.level_pixels(ss) == 16 and A.level_pixels(ss) == 16
    ss[0, 0] = 5                                   # in-place change -> new version -> recomputed
    assert A.level_pixels(ss) == 24
    other = torch.tensor([[1, 1], [2, 2]])
    assert A.level_pixels(other) == 5
    del ss
    fresh = torch.tensor([[7, 1], [1, 1]])         # (may reuse the id of the dead tensor: the weak reference catches it)
    assert A.level_pixels(fresh) == 8
    mod = A.MSDeformAttn(d_model=32, n_levels=2, n_heads=2, n_points=2)
    q, src = torch.randn(1, 5, 32), torch.randn(1, 9, 32)
    import pytest
    with pytest.raises(AssertionError):            # the module's check still fires (ms_deform_attn.py:100)
        mod(q, torch.rand(1, 5, 2, 2), src, fresh, torch.tensor([0, 7]), None)


def test_level_geometry_facts_on_the_host():
    """Host side of the MSDA geometry hint (ms_deform_attn.shape_facts / known_geometry / nested_maps): exact 2x pyramids,
    ceil- / floor-divided maps (what the pyramid-item kernel's nested instantiation serves), everything else; the answer is
    remembered per tensor object and version, only for the encoder's query count, and dropped on an in-place change.
    The constants equal the header's."""
    import re
    import torch
    from visionllm_amd import ms_deform_attn as A
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vllm_hip.h")).read()
    for name, val in (("UNKNOWN", A.GEO_UNKNOWN), ("PYRAMID", A.GEO_PYRAMID), ("GENERAL", A.GEO_GENERAL), ("NESTED", A.GEO_NESTED)):
        assert int(re.search(rf"#define VLLM_GEO_{name} (\d+)", hdr).group(1)) == val
    assert A.nested_maps([(168, 168), (84, 84), (42, 42), (21, 21)])
    assert A.nested_maps([(100, 167), (50, 84), (25, 42), (13, 21)])          # ceil-divided
    assert A.nested_maps([(51, 83), (25, 41), (12, 20), (6, 10)])             # floor-divided
    assert A.nested_maps([(72, 64)]) and not A.nested_maps([])
    assert not A.nested_maps([(61, 83), (31, 42), (9, 5)])                    # not halves
    assert not A.nested_maps([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)])  # five levels
    assert not A.nested_maps([(64, 64), (33, 32)])                            # 33 > ceil(64 / 2)
    for shapes, geo in (([(72, 64), (36, 32)], A.GEO_PYRAMID), ([(50, 83), (25, 42)], A.GEO_NESTED), ([(50, 83), (20, 42)], A.GEO_GENERAL)):
        t = torch.tensor(shapes, dtype=torch.int64)
        lq = sum(h * w for h, w in shapes)
        assert A.known_geometry(t, lq) == A.GEO_UNKNOWN
        assert A.shape_facts(t) == (lq, geo)
        assert A.known_geometry(t, lq) == geo
        assert A.known_geometry(t, lq - 1) == A.GEO_GENERAL                   # the pyramid-item kernel serves the encoder's queries only
        t.mul_(1)
        assert A.known_geometry(t, lq) == A.GEO_UNKNOWN
