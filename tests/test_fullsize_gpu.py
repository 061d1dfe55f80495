"""BASELINE.json configs[1] and configs[2] AT THE SIZES THEY STATE, end to end through the native
path, against fixtures made by running the reference's own classes on the host (oracle/gen_golden_fullsize.py):

  configs[1]  ViT-L/14-336, 24 layers, 32 tiles + mlp2x_gelu projector (the shape bench.py times, 32 instead of 40 tiles)
  configs[2]  InternViT-6B, 48 layers, the 5 tiles of one 1336^2 image + pixel-shuffle + internvl_mlp 12800 -> 4096 -> 4096

Weights / pixels are the deterministic hash tensors of oracle/detweights.py, regenerated here ON THE DEVICE (bit-identical to
the ones the fixture was made with: the fixture's CRC is checked).  The tolerance contract is DESIGN.md section 5's:
  * a bf16 tensor cannot be within 1e-3 of an fp32 reference element-wise (half an ulp is 2^-9 = 2e-3 relative), so `north_star`'s
    "1e-3 bf16 tolerance" is met where it can be (MSDA fp32: 4e-6; single kernels: <= 1 bf16 ulp) and, for whole encoders, the bar is
    "not further from the fp32 truth than the reference's OWN bf16 arithmetic": relative rms <= 1.25 x the bf16 reference run's
    (per hidden state and for the visual tokens), worst element <= 2 x its worst element;
  * element-wise comparisons use the fixture's strided subsample of 5 hidden states; WHOLE tensors are covered by digests of EVERY
    hidden state and of the tokens (row and column projections on hash vectors: every element enters both), under the same kind of
    bar: <= 1.25 x the digest error of the reference's bf16 run (`_check_digests`);
  * the measured numbers of every config are written to gpurun_out/parity_contract.jsonl (copied to profiles/ per round).
"""
import ast
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import detweights as DW
from visionllm_amd.bridge import build_vl_bridge
from visionllm_amd.clip_vit import CLIPVisionModel
from visionllm_amd.intern_vit import InternVisionConfig, InternVisionModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(row):
    path = os.path.join(ROOT, "gpurun_out", "parity_contract.jsonl")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        f.write(json.dumps(row) + "\n")


def _sub(t, ts, cs):
    return t[:, ::ts, ::cs].float().cpu()


def _digests(t, tag):
    """The whole-tensor digests of oracle/gen_golden_fullsize.py::digests, evaluated on the device in fp32: every row projected on a
    hash vector over the channels, every channel on a hash vector over the rows -- every element enters both."""
    t = t.float()
    u = DW.hash_uniform(f"fullsize.{tag}.u", (t.shape[2],), 1.0, 0.0, device=t.device).float()
    v = DW.hash_uniform(f"fullsize.{tag}.v", (t.shape[1],), 1.0, 0.0, device=t.device).float()
    return torch.matmul(t, u).cpu(), torch.einsum("nsc,s->nc", t, v).cpu()


def _check_digests(tag, g, key, kind, tensors):
    """WHOLE tensors: the digests of EVERY hidden state / of the visual tokens against the fp32 reference run's,
    with the reference's own bf16 run as the yardstick (same form as the element contract: <= 1.25 x its digest error, floor 3e-3)."""
    worst = dict(config=tag, tensor=f"{kind}: whole-tensor digests", n_tensors=len(tensors), row_ratio=0.0, col_ratio=0.0)
    for i, t in enumerate(tensors):
        ours = _digests(t, key)
        for j, name in enumerate(("rowproj", "colproj")):
            ref = torch.from_numpy(g[f"{key}{i}.{name}"].astype(np.float32)) * float(g[f"{key}{i}.{name}.scale"])
            assert ours[j].shape == ref.shape, (ours[j].shape, ref.shape)
            rel = ((ours[j] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            lo = float(g[f"{key}_digest.lo_rel_rms_{name[:3]}"][i])
            bound = max(1.25 * lo, 3e-3)
            rk = f"{name[:3]}_ratio"
            if rel / bound > worst[rk]:
                worst.update({rk: rel / bound, f"{name[:3]}_worst": dict(index=i, rel_rms=rel, ref_bf16_rel_rms=lo)})
            assert rel <= bound, dict(config=tag, tensor=f"{kind}[{i}].{name}", rel_rms=rel, ref_bf16_rel_rms=lo)
    _record(worst)
    print(json.dumps(worst))


def _check(tag, g, hidden_states, tokens):
    ts, cs, tts, tcs = (int(v) for v in g["strides"])
    n_states = len(g["hs_stats.rms"])
    assert len(hidden_states) == n_states, (len(hidden_states), n_states)
    _check_digests(tag, g, "hs", "hidden_state", list(hidden_states))
    _check_digests(tag, g, "tok", "visual tokens", [tokens])
    rows = []
    for i in (int(v) for v in g["kept"]):
        ref = torch.from_numpy(g[f"hs{i}"].astype(np.float32))
        ours = _sub(hidden_states[i], ts, cs)
        assert ours.shape == ref.shape, (ours.shape, ref.shape)
        rms_ref = float(g["hs_stats.rms"][i])
        rel = ((ours - ref).pow(2).mean().sqrt() / rms_ref).item()
        mx = (ours - ref).abs().max().item()
        lo_rel, lo_max = float(g["hs_stats.lo_rel_rms"][i]), float(g["hs_stats.lo_max_abs"][i])
        rows.append(dict(config=tag, tensor=f"hidden_state[{i}]", rel_rms=rel, max_abs=mx, ref_bf16_rel_rms=lo_rel,
                         ref_bf16_max_abs=lo_max, ref_rms=rms_ref, ref_absmax=float(g["hs_stats.absmax"][i]),
                         meets_1e3_abs=bool(mx <= 1e-3)))
        # (the subsample's relative rms against the full tensor's bf16-run figure: 75 K+ elements, the sampling error is < 1 %)
        assert rel <= max(1.25 * lo_rel, 2e-3), rows[-1]
        assert mx <= max(2.0 * lo_max, 2e-2 * float(g["hs_stats.absmax"][i])), rows[-1]
    ref = torch.from_numpy(g["tokens"].astype(np.float32))
    lo = torch.from_numpy(g["tokens_lo"].astype(np.float32))
    ours = _sub(tokens, tts, tcs)
    assert ours.shape == ref.shape, (ours.shape, ref.shape)
    rms_ref = float(g["tok_stats.rms"][0])
    rel = ((ours - ref).pow(2).mean().sqrt() / rms_ref).item()
    mx = (ours - ref).abs().max().item()
    lo_rel, lo_max = float(g["tok_stats.lo_rel_rms"][0]), float(g["tok_stats.lo_max_abs"][0])
    lo_rel_sub = ((lo - ref).pow(2).mean().sqrt() / rms_ref).item()
    rows.append(dict(config=tag, tensor="visual tokens", rel_rms=rel, max_abs=mx, ref_bf16_rel_rms=lo_rel, ref_bf16_max_abs=lo_max,
                     ref_bf16_rel_rms_on_subsample=lo_rel_sub, ref_rms=rms_ref, ref_absmax=float(g["tok_stats.absmax"][0]),
                     meets_1e3_abs=bool(mx <= 1e-3)))
    for row in rows:
        _record(row)
        print(json.dumps(row))
    assert rel <= max(1.25 * lo_rel, 2e-3), rows[-1]
    assert mx <= max(2.0 * lo_max, 2e-2 * float(g["tok_stats.absmax"][0])), rows[-1]


def test_cfg2_vitl14_336_batch32_full_depth_vs_reference_run():
    """BASELINE configs[1] as stated: 32 tiles x 24 layers + projector, vs transformers.CLIPVisionModel run on the host."""
    from transformers import CLIPVisionConfig
    g = load_golden("fullsize_cfg2.npz")
    cfgd = ast.literal_eval(str(g["cfg"]))
    n = int(g["n_tiles"])
    with torch.device(DEV):
        model = CLIPVisionModel(CLIPVisionConfig(**cfgd)).to(torch.bfloat16)
        br = build_vl_bridge("mlp2x_gelu", 1024, 4096, use_pixelshuffle=False).to(torch.bfloat16)
    assert DW.fill_module_(model, DW.clip_param) == int(g["crc_encoder"]), "weight generator differs from the fixture's"
    assert DW.fill_module_(br, DW.bridge_param) == int(g["crc_bridge"])
    model = model.eval().requires_grad_(False)
    br = br.eval().requires_grad_(False)
    x = DW.pixels("cfg2.pixels", n, cfgd["image_size"], device=DEV)
    out = model(x, output_hidden_states=True)
    tok = br.project_hidden_state(out.hidden_states[int(g["select"])], False)
    torch.cuda.synchronize()
    assert tok.shape == (n, 576, 4096)
    _check("configs[1] ViT-L/14-336 x 32 tiles x 24 layers + mlp2x_gelu", g, out.hidden_states, tok)


def test_cfg3_internvit6b_5tiles_full_depth_plus_projector_vs_reference_run():
    """BASELINE configs[2] as stated: the 5 tiles of a 1336^2 image x 48 layers + pixel-shuffle + internvl_mlp at 12800 -> 4096,
    vs the reference's InternVisionModel run on the host."""
    g = load_golden("fullsize_cfg3.npz")
    cfgd = ast.literal_eval(str(g["cfg"]))
    n = int(g["n_tiles"])
    with torch.device(DEV):
        model = InternVisionModel(InternVisionConfig(**cfgd)).to(torch.bfloat16)
        br = build_vl_bridge("internvl_mlp", 3200, 4096, use_pixelshuffle=True).to(torch.bfloat16)
    assert DW.fill_module_(model, DW.intern_vit_param) == int(g["crc_encoder"]), "weight generator differs from the fixture's"
    assert DW.fill_module_(br, DW.bridge_param) == int(g["crc_bridge"])
    model = model.eval().requires_grad_(False)
    br = br.eval().requires_grad_(False)
    x = DW.pixels("cfg3.pixels", n, cfgd["image_size"], device=DEV)
    out = model(x, output_hidden_states=True, return_dict=True)
    tok = br.project_hidden_state(out.hidden_states[int(g["select"])], True)
    torch.cuda.synchronize()
    assert tok.shape == (n, 256, 4096)
    _check("configs[2] InternViT-6B x 5 tiles x 48 layers + pixel-shuffle + internvl_mlp", g, out.hidden_states, tok)


@pytest.mark.parametrize("kind,cin,ps", [("mlp2x_gelu", 1024, False), ("internvl_mlp", 3200, True), ("linear", 1024, False)])
def test_bridge_real_width_vs_fp32(kind, cin, ps):
    """The projector at the widths the two released configurations use: LayerNorm(12800) + K = 12800 GEMM +
    pixel-shuffle for internvl_mlp, 1024 -> 4096 -> 4096 for mlp2x_gelu -- against the same modules evaluated in fp32 by torch on
    the device (a plain fp32 reference of the reference's nn.Sequential, modeling_visionllmv2.py:162-182) and in bf16 (its own
    arithmetic at the deployed precision)."""
    from oracle import vit as V
    n, hw = 4, (24 if cin == 1024 else 32)
    with torch.device(DEV):
        br = build_vl_bridge(kind, cin, 4096, use_pixelshuffle=ps).to(torch.bfloat16)
    DW.fill_module_(br, DW.bridge_param)
    br = br.eval().requires_grad_(False)
    hidden = DW.hash_uniform(f"bridge.{kind}.hidden", (n, 1 + hw * hw, cin), 1.5, 0.0, device=DEV)
    sd32 = {k: v.detach().float() for k, v in br.state_dict().items()}
    feats = V.select_features([hidden.float(), hidden.float()], -2, ps)
    ref = V.bridge_forward(sd32, kind, feats)                                                    # fp32 on the device
    lo = V.bridge_forward({k: v.to(torch.bfloat16) for k, v in sd32.items()}, kind, feats.to(torch.bfloat16)).float()   # bf16 torch ops
    out = br.project_hidden_state(hidden, ps).float()
    assert out.shape == ref.shape
    rms = ref.pow(2).mean().sqrt()
    rel, lo_rel = ((out - ref).pow(2).mean().sqrt() / rms).item(), ((lo - ref).pow(2).mean().sqrt() / rms).item()
    mx, lo_max = (out - ref).abs().max().item(), (lo - ref).abs().max().item()
    row = dict(config=f"projector {kind} {cin}{' x4 (pixel-shuffle)' if ps else ''} -> 4096", tensor="visual tokens", rel_rms=rel,
               max_abs=mx, ref_bf16_rel_rms=lo_rel, ref_bf16_max_abs=lo_max, ref_rms=rms.item(), ref_absmax=ref.abs().max().item(),
               meets_1e3_abs=bool(mx <= 1e-3))
    _record(row)
    print(json.dumps(row))
    assert rel <= max(1.1 * lo_rel, 2e-3), row
    assert mx <= max(2.0 * lo_max, 1e-2 * ref.abs().max().item()), row
