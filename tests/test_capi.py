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
    assert L.vllm_abi_version() == _lib.header_abi_version() == 2   # (2: round-4 bump for the round-3 struct changes)


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


def test_host_side_queries_of_round_5_need_no_gpu():
    """The two pure host queries added in round 5: the splice workspace size, and whether the MSDA backward writes every per-point
    gradient itself (decided from sizes / alignment alone; no kernel runs)."""
    import ctypes
    from visionllm_amd import _lib
    L = _lib.lib()
    assert L.vllm_splice_workspace_ints(8, 4096, 40) == 4 + 8 * 4096 + 40
    assert L.vllm_splice_workspace_ints(600, 16, 7) == 4 + 600 * 16 + 7 + 600          # > 512 samples: the tiles per sample travel through it
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # decoder cross-attention shape (Lq != S) and a tiny encoder shape (below the tiled kernels' size threshold): the caller zero-fills
    assert L.vllm_msda_backward_f32_writes_point_grads(p, p, p, p, p, 2, 5000, 8, 32, 4, 900, 4) == 0
    assert L.vllm_msda_backward_f32_writes_point_grads(p, p, p, p, p, 2, 64, 8, 32, 4, 64, 4) == 0
    # the encoder self-attention shape of BASELINE cfg 4 on aligned buffers: the matrix-core kernel writes them all
    ptr = ctypes.c_void_p(1 << 20)
    assert L.vllm_msda_backward_f32_writes_point_grads(ptr, ptr, ptr, ptr, ptr, 8, 37485, 8, 32, 4, 37485, 4) == 1
    odd = ctypes.c_void_p((1 << 20) + 4)                                              # grad_loc only 4-byte aligned: another kernel runs
    assert L.vllm_msda_backward_f32_writes_point_grads(ptr, ptr, ptr, ptr, odd, 8, 37485, 8, 32, 4, 37485, 4) == 0


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


def test_folded_norm_operands_reproduce_norm_then_linear():
    """Host logic of the folded norms (vit_common.fold_norm_into_linear): with exact row statistics,
    r (x W'^T) - r mean colsum + bias' equals linear(LayerNorm(x)) for the bf16 weights in use (fp64 here), and the RMSNorm form
    has neither column sums nor a shifted bias.  The GPU tests check the kernels against this expression."""
    import torch
    from visionllm_amd.vit_common import fold_norm_into_linear, norm_folding_applies
    torch.manual_seed(0)
    C, N, M, eps = 96, 40, 17, 1e-5
    x = (torch.randn(M, C) * 2 + 0.7).to(torch.bfloat16).double()
    w = (torch.randn(N, C) / C ** 0.5).to(torch.bfloat16)
    b = torch.randn(N).to(torch.bfloat16)
    gamma = (1 + 0.2 * torch.randn(C)).to(torch.bfloat16)
    beta = (0.1 * torch.randn(C)).to(torch.bfloat16)
    w_ln, colsum, bias_ln = fold_norm_into_linear(w, b, gamma, beta, layernorm=True)
    assert w_ln.dtype == torch.bfloat16 and colsum.dtype == torch.float32 and bias_ln.dtype == torch.float32
    mu = x.mean(1, keepdim=True)
    r = torch.rsqrt(((x - mu) ** 2).mean(1, keepdim=True) + eps)
    folded = r * (x @ w_ln.double().t()) - r * mu * colsum.double()[None, :] + bias_ln.double()[None, :]
    # the same weights the folded GEMM multiplies with: gamma W rounded to bf16 (the reference rounds gamma * xhat instead)
    direct = ((x - mu) * r) @ w_ln.double().t() + (beta.double() @ w.double().t() + b.double())[None, :]
    assert (folded - direct).abs().max().item() < 1e-5
    w_r, colsum_r, bias_r = fold_norm_into_linear(w, b, gamma, None, layernorm=False)
    assert colsum_r is None and torch.equal(bias_r, b.float()) and torch.equal(w_r, w_ln)
    r2 = torch.rsqrt((x * x).mean(1, keepdim=True) + eps)
    assert ((r2 * (x @ w_r.double().t()) + bias_r.double()) - ((x * r2) @ w_r.double().t() + b.double())).abs().max().item() < 1e-9
    assert norm_folding_applies(1024, 4096) and not norm_folding_applies(3200, 12800)


def test_tile_statistics_combine_to_the_row_statistics():
    """Chan's update as the GEMM epilogues use it: per-tile {mean, M2} of 256 (last tile: fewer) values combine, in a fixed order and
    with the launcher's constants (n_b / n, n_a n_b / n), to the row's mean and centred second moment; the equal-count form
    (pairs of 8, 16, 32, 64, 128 values) is the producer's in-register reduction."""
    import numpy as np
    rng = np.random.default_rng(1)
    for cols in (1024, 960, 800):
        x = rng.standard_normal((5, cols)) * 3 + 40.0                      # a mean far from zero: the one-pass form would cancel
        cnt = mean = m2 = 0.0
        mean = np.zeros(5); m2 = np.zeros(5); cnt = 0.0
        for s in range(4):
            blk = x[:, s * 256:min(cols, (s + 1) * 256)]
            nb = blk.shape[1]
            bm = blk.mean(1); bq = ((blk - bm[:, None]) ** 2).sum(1)
            tot = cnt + nb
            d = bm - mean
            mean = mean + d * (nb / tot)
            m2 = m2 + bq + d * d * (cnt * nb / tot)
            cnt = tot
        np.testing.assert_allclose(mean, x.mean(1), rtol=1e-13)
        np.testing.assert_allclose(m2, ((x - x.mean(1, keepdims=True)) ** 2).sum(1), rtol=1e-12)
    x = rng.standard_normal((3, 256)) + 7.0
    parts = [(x[:, i:i + 8].mean(1), ((x[:, i:i + 8] - x[:, i:i + 8].mean(1, keepdims=True)) ** 2).sum(1)) for i in range(0, 256, 8)]
    n = 8
    while len(parts) > 1:
        nxt = []
        for (ma, qa), (mb, qb) in zip(parts[0::2], parts[1::2]):
            d = mb - ma
            nxt.append((ma + 0.5 * d, qa + qb + d * d * (n / 2)))
        parts, n = nxt, 2 * n
    np.testing.assert_allclose(parts[0][0], x.mean(1), rtol=1e-13)
    np.testing.assert_allclose(parts[0][1], ((x - x.mean(1, keepdims=True)) ** 2).sum(1), rtol=1e-12)


def test_vit_layer_descriptor_matches_the_header():
    """The ctypes mirror of VllmVitLayer carries the six folded-norm operands at the END of the struct (a round-2 caller's layout is
    a prefix of it), in the header's order."""
    names = [f[0] for f in _lib.VllmVitLayer._fields_]
    assert names[-6:] == ["qkv_w_ln", "qkv_colsum", "qkv_bias_ln", "fc1_w_ln", "fc1_colsum", "fc1_bias_ln"]
    assert names[:4] == ["norm1_w", "norm1_b", "qkv_w", "qkv_b"]
    assert ctypes.sizeof(_lib.VllmVitLayer) == len(names) * ctypes.sizeof(ctypes.c_void_p)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vllm_hip.h")).read()
    body = hdr[hdr.index("typedef struct VllmVitLayer {"):hdr.index("} VllmVitLayer;")]
    order = [body.index(n) for n in names]
    assert order == sorted(order), "field order differs from include/vllm_hip.h"


def test_persistent_gemm_keeps_in_flight_registers_out_of_scratch(tmp_path):
    """gemm256p.hip loads the residual tile with inline buffer loads the compiler does not know are loads: if register pressure
    ever made it spill one of those destination registers it would store the register BEFORE the data has arrived (no wait is
    inserted for an instruction it cannot see).  The residual instantiations must therefore compile without vector spills --
    checked here on the code object's metadata (cross-compiles without a GPU)."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "visionllm_amd", "csrc", "gemm256p.hip")
    out = tmp_path / "gemm256p.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-S",
                    "--cuda-device-only", src, "-o", str(out)], check=True, cwd=os.path.dirname(src), capture_output=True, timeout=600)
    text = out.read_text()
    names = re.findall(r"\.name:\s+(\S*gemm256p_kernel\S*)", text)
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    assert names and len(names) == len(spills)
    residual = [(n, int(s)) for n, s in zip(names, spills) if "gemm256p_kernelILi3E" in n]          # EPI_RESIDUAL = 3
    assert len(residual) == 6, residual                                                             # MT 3 / 4 x {no statistics, pairs, wide (round 5)}
    plain = [(n, s) for n, s in residual if n.endswith("Lb0ELi0ELb0EEEvNS_8GemmArgsE")]
    assert len(plain) == 2, plain
    assert all(s == 0 for _, s in plain), plain
    # the statistics variant at MT = 4 spills ONE scalar set-up value at the kernel's top (reloaded at its very end): nothing in flight
    assert all(s <= 1 for _, s in residual), residual
    others = [(n, int(s)) for n, s in zip(names, spills) if "gemm256p_kernelILi3E" not in n]
    assert all(s == 0 for _, s in others), others


def test_attention_lane_swaps_read_both_results(tmp_path):
    """v_permlane32_swap writes BOTH its registers; this toolchain has been seen to fold the builtin's two results into one
    (profiles/r03_permlane_swap_codegen.txt: r[0] + r[1] compiled to v + v in a small kernel).  The attention kernels combine the
    two key halves of a row with it (attn_common.hpp: halves_max / halves_sum): every swap in their code objects must be followed
    by a read of its second register before that register is overwritten."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("attn.hip",):   # (attn2.hip left the library in round 4: tools/experiments/)
        src = os.path.join(root, "visionllm_amd", "csrc", name)
        out = tmp_path / (name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-S",
                        "--cuda-device-only", src, "-o", str(out)], check=True, cwd=os.path.dirname(src), capture_output=True, timeout=900)
        lines = [ln for ln in out.read_text().splitlines() if ln.strip() and not ln.strip().startswith(";")]
        swaps = 0
        for i, ln in enumerate(lines):
            m = re.search(r"v_permlane32_swap_b32\S*\s+(v\d+), (v\d+)", ln)
            if not m:
                continue
            swaps += 1
            second = m.group(2)
            verdict = None
            for nxt in lines[i + 1:i + 40]:
                ops = nxt.split(None, 1)
                if len(ops) < 2 or not ops[0].startswith(("v_", "ds_", "global_", "buffer_", "scratch_")):
                    continue
                dst, _, srcs = ops[1].partition(",")
                if re.search(r"\b" + second + r"\b", srcs):
                    verdict = "read"
                    break
                if re.fullmatch(second, dst.strip()):
                    verdict = "overwritten"
                    break
            assert verdict == "read", f"{name}: second result of the swap at instruction {i} ({ln.strip()}) is {verdict}"
        assert swaps > 0, name


def test_every_runtime_option_is_documented_in_the_header():
    """vllm_set_option's names (csrc/runtime.cpp) are part of the boundary: each must be described in include/vllm_hip.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "visionllm_amd", "csrc", "runtime.cpp")).read()
    hdr = open(os.path.join(root, "include", "vllm_hip.h")).read()
    names = sorted(set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', src)))
    assert len(names) >= 8, names
    missing = [n for n in names if f'"{n}"' not in hdr]
    assert not missing, f"options without a description in include/vllm_hip.h: {missing}"
