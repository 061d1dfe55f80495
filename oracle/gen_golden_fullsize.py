#!/usr/bin/env python3
"""ORACLE -- test infrastructure only.  FULL-SIZE parity fixtures, made by RUNNING THE REFERENCE CLASSES at the sizes
BASELINE.json states:

  * configs[1]  ViT-L/14-336, 24 layers, batch 32 tiles  (transformers.CLIPVisionModel, the class the reference instantiates at
                modeling_visionllmv2.py:135) + the mlp2x_gelu vl_bridge 1024 -> 4096 -> 4096 built as :174-182 builds it
  * configs[2]  InternViT-6B (hidden 3200, 25 heads, 48 layers), the 5 tiles (448^2) of ONE 1336^2 image, through the reference's
                own InternVisionModel (modeling_intern_vit.py, imported with the DropPath stub of gen_golden.py) + pixel_shuffle
                (:381-392, AST-extracted) + the internvl_mlp vl_bridge LN(12800) -> 12800 -> 4096 -> 4096 (:166-172)

Run in the build container only (needs /root/reference; ~25 GB of RAM and ~10 minutes for configs[2]):

    python oracle/gen_golden_fullsize.py [cfg2] [cfg3]

Weights and pixels are the deterministic hash tensors of oracle/detweights.py (the GPU test regenerates them bit for bit on
the device; a fixture cannot carry 5.9 G parameters).  Each reference model runs twice on the host: in fp32 (the truth) and in
bf16 (the reference's own arithmetic at the precision it is deployed in -- the yardstick of DESIGN section 5).  A fixture holds
  * a strided SUBSAMPLE (tokens ::ts, channels ::cs; fp16 storage) of selected fp32 hidden states and of the visual tokens,
  * full-tensor statistics of EVERY hidden state: rms / absmax of the fp32 run, relative rms and max |error| of the bf16 run,
  * the same for the visual tokens, and the CRC of the generated weights,
  * WHOLE-TENSOR digests of every hidden state and of the visual tokens (round 5; VERDICT r4 weak #1 "a strided subsample, not whole
    tensors"): the projection of every row on a fixed hash vector over the channels (`rowproj`, [N, S]) and of every channel on a
    fixed hash vector over the rows (`colproj`, [N, C]) -- every element of the tensor enters both with a different weight -- of the
    fp32 run (stored as fp16 x a power-of-two scale), plus the relative rms error of the bf16 run's digests (the yardstick).
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import detweights as DW          # noqa: E402
from oracle import gen_golden as GG          # noqa: E402

OUT = GG.OUT
REF = GG.REF


def sub(t, ts, cs):
    return t[:, ::ts, ::cs].contiguous()


def stats(ref, lo=None):
    r = ref.float()
    d = dict(rms=r.pow(2).mean().sqrt().item(), absmax=r.abs().max().item())
    if lo is not None:
        e = lo.float() - r
        d.update(lo_rel_rms=(e.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item(), lo_max_abs=e.abs().max().item())
    return d


def digests(t, tag):
    """[N, S] and [N, C] projections of a [N, S, C] tensor on the hash vectors the GPU test regenerates (oracle/detweights.py)."""
    t = t.float()
    u = DW.hash_uniform(f"fullsize.{tag}.u", (t.shape[2],), 1.0, 0.0).float()
    v = DW.hash_uniform(f"fullsize.{tag}.v", (t.shape[1],), 1.0, 0.0).float()
    return torch.matmul(t, u), torch.einsum("nsc,s->nc", t, v)


def digest_rows(save, key, refs, los, tag):
    rel_r, rel_c = [], []
    for i, (r, l) in enumerate(zip(refs, los)):
        pr, pc = digests(r, tag)
        lr, lc = digests(l, tag)
        for name, d in (("rowproj", pr), ("colproj", pc)):   # fp16 storage at a per-tensor power-of-two scale: 5e-4 relative, the errors
            sc = 2.0 ** np.ceil(np.log2(max(d.abs().max().item(), 1e-30) / 16384.0))   # measured on them are 2e-3 ... 2e-2
            save[f"{key}{i}.{name}"] = (d / sc).numpy().astype(np.float16)
            save[f"{key}{i}.{name}.scale"] = np.array(sc)
        rel_r.append(((lr - pr).pow(2).mean().sqrt() / pr.pow(2).mean().sqrt()).item())
        rel_c.append(((lc - pc).pow(2).mean().sqrt() / pc.pow(2).mean().sqrt()).item())
    save[f"{key}_digest.lo_rel_rms_row"] = np.array(rel_r)
    save[f"{key}_digest.lo_rel_rms_col"] = np.array(rel_c)


def pack(save, key, rows):
    for k in rows[0]:
        save[f"{key}.{k}"] = np.array([row[k] for row in rows], dtype=np.float64)


def run_case(tag, model, bridge_fn, make_bridge, x, keep, ts, cs, tts, tcs, select):
    """model: reference encoder (fp32, eval); bridge_fn(list of hidden states, bridge module) -> tokens."""
    save = {}
    t0 = time.time()
    with torch.no_grad():
        hs = model(pixel_values=x.float(), output_hidden_states=True, return_dict=True).hidden_states
        hs = [h.float() for h in hs]
        br = make_bridge().eval().float()
        crc_b = DW.fill_module_(br, DW.bridge_param)
        tok = bridge_fn(hs, br).float()
    t1 = time.time()
    print(f"{tag}: fp32 reference run {t1 - t0:.1f}s, {len(hs)} hidden states {tuple(hs[0].shape)}, tokens {tuple(tok.shape)}", flush=True)
    for i in keep:
        save[f"hs{i % len(hs)}"] = sub(hs[i], ts, cs).numpy().astype(np.float16)
    save["tokens"] = sub(tok, tts, tcs).numpy().astype(np.float16)
    # the reference's own arithmetic in bf16
    model = model.to(torch.bfloat16)
    br = br.to(torch.bfloat16)
    with torch.no_grad():
        hl = model(pixel_values=x.to(torch.bfloat16), output_hidden_states=True, return_dict=True).hidden_states
        tl = bridge_fn(list(hl), br)
    t2 = time.time()
    print(f"{tag}: bf16 reference run {t2 - t1:.1f}s", flush=True)
    pack(save, "hs_stats", [stats(r, l) for r, l in zip(hs, hl)])
    pack(save, "tok_stats", [stats(tok, tl)])
    digest_rows(save, "hs", hs, hl, "hs")
    digest_rows(save, "tok", [tok], [tl], "tok")
    print(f"{tag}: bf16-run digest errors: rows {save['hs_digest.lo_rel_rms_row'][[0, len(hs) // 2, -1]]}, columns "
          f"{save['hs_digest.lo_rel_rms_col'][[0, len(hs) // 2, -1]]}, tokens {save['tok_digest.lo_rel_rms_row']} / "
          f"{save['tok_digest.lo_rel_rms_col']}", flush=True)
    save["tokens_lo"] = sub(tl.float(), tts, tcs).numpy().astype(np.float16)
    save["select"] = np.array(select)
    save["strides"] = np.array([ts, cs, tts, tcs], dtype=np.int64)
    save["kept"] = np.array([i % len(hs) for i in keep], dtype=np.int64)
    save["crc_bridge"] = np.array(crc_b, dtype=np.int64)
    save["seconds"] = np.array([t1 - t0, t2 - t1])
    st = save["hs_stats.lo_rel_rms"]
    print(f"{tag}: bf16-run relative rms vs fp32: first {st[0]:.3g} mid {st[len(st) // 2]:.3g} last {st[-1]:.3g}; tokens "
          f"{save['tok_stats.lo_rel_rms'][0]:.3g} (max |err| {save['tok_stats.lo_max_abs'][0]:.3g}, absmax {save['tok_stats.absmax'][0]:.3g})", flush=True)
    return save


def gen_cfg2():
    """BASELINE configs[1]: ViT-L/14 bf16, batch 32 at 336x336."""
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfgd = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24, image_size=336,
                patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    cfg = CLIPVisionConfig(**cfgd)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    model = CLIPVisionModel(cfg).eval().float()
    crc = DW.fill_module_(model, DW.clip_param)
    x = DW.pixels("cfg2.pixels", 32, 336)

    def make_bridge():   # modeling_visionllmv2.py:174-182 (mlp2x_gelu)
        return nn.Sequential(nn.Linear(1024, 4096), nn.GELU(), nn.Linear(4096, 4096))

    def bridge_fn(hs, br):   # :569-579 without pixel shuffle
        return br(hs[-2][:, 1:].to(next(br.parameters()).dtype))

    save = run_case("cfg2", model, bridge_fn, make_bridge, x, keep=[0, 1, 12, -2, -1], ts=16, cs=16, tts=16, tcs=64, select=-2)
    save.update(cfg=np.array(repr(cfgd)), n_tiles=np.array(32), crc_encoder=np.array(crc, dtype=np.int64),
                transformers_version=np.array(transformers.__version__), bridge=np.array("mlp2x_gelu"))
    np.savez_compressed(os.path.join(OUT, "fullsize_cfg2.npz"), **save)


def gen_cfg3():
    """BASELINE configs[2]: InternViT-6B encoder + projector, one 1336x1336 image = 5 tiles of 448x448."""
    Cfg, Model = GG.load_intern_vit()
    cfgd = dict(hidden_size=3200, num_attention_heads=25, intermediate_size=12800, num_hidden_layers=48, image_size=448,
                patch_size=14, qk_normalization=True, qkv_bias=False, hidden_act="gelu", layer_norm_eps=1e-6)
    t0 = time.time()
    # (construct without running the default initialisers over 5.9 G parameters: every tensor is overwritten below)
    saved = {n: getattr(nn.init, n) for n in ("normal_", "trunc_normal_", "kaiming_uniform_", "uniform_", "zeros_", "ones_")}
    try:
        for n in saved:
            setattr(nn.init, n, lambda t, *a, **k: t)
        model = Model(Cfg(use_flash_attn=False, **cfgd)).eval().float()
    finally:
        for n, f in saved.items():
            setattr(nn.init, n, f)
    crc = DW.fill_module_(model, DW.intern_vit_param)
    print(f"cfg3: model built + filled in {time.time() - t0:.1f}s", flush=True)
    x = DW.pixels("cfg3.pixels", 5, 448)
    glb = {"torch": torch, "int": int}
    ps = GG.ast_extract(f"{REF}/visionllmv2/model/modeling_visionllmv2.py", ["pixel_shuffle"], glb)["pixel_shuffle"]

    def make_bridge():   # modeling_visionllmv2.py:166-172 (internvl_mlp, downsample_ratio 0.5 -> 4 x 3200 = 12800 inputs)
        return nn.Sequential(nn.LayerNorm(12800), nn.Linear(12800, 4096), nn.GELU(), nn.Linear(4096, 4096))

    def bridge_fn(hs, br):   # :569-579 with pixel shuffle
        f = hs[-2][:, 1:].to(next(br.parameters()).dtype)
        h = w = int(f.shape[1] ** 0.5)
        f = ps(None, f.reshape(f.shape[0], h, w, -1), scale_factor=0.5)
        return br(f.reshape(f.shape[0], -1, f.shape[-1]))

    save = run_case("cfg3", model, bridge_fn, make_bridge, x, keep=[0, 1, 24, -2, -1], ts=16, cs=16, tts=4, tcs=32, select=-2)
    save.update(cfg=np.array(repr(cfgd)), n_tiles=np.array(5), crc_encoder=np.array(crc, dtype=np.int64), bridge=np.array("internvl_mlp"))
    np.savez_compressed(os.path.join(OUT, "fullsize_cfg3.npz"), **save)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["cfg2", "cfg3"]
    os.makedirs(OUT, exist_ok=True)
    if "cfg2" in which:
        gen_cfg2()
    if "cfg3" in which:
        gen_cfg3()
