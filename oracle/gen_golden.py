#!/usr/bin/env python3
"""ORACLE -- test infrastructure only.  Generates tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python oracle/gen_golden.py

What is imported / executed from the reference (read-only, nothing is copied into the repo):
  * mmcv/mmcv/ops/multi_scale_deform_attn.py:100-159  ``multi_scale_deformable_attn_pytorch``
    -- AST-extracted (``import mmcv`` needs the compiled ``_ext``), exec'd with {torch, F}.
  * visionllmv2/model/internvit/{modeling_intern_vit,configuration_intern_vit}.py -- imported with a
    6-line identity ``timm.models.layers.DropPath`` stub (``_naive_attn`` + python ``InternRMSNorm`` path).
  * visionllmv2/model/modeling_visionllmv2.py:381-392 ``pixel_shuffle`` -- AST-extracted method.
  * visionllmv2/mm_utils.py:23-77 ``find_closest_aspect_ratio`` / ``dynamic_preprocess`` -- AST-extracted,
    run with a 2-method fake PIL image (only the tile count / grid is recorded).
  * transformers.CLIPVisionModel (third-party class the reference instantiates at
    modeling_visionllmv2.py:135; local transformers version recorded in the fixture).

The MSDA known-answer inputs are the reference's own: mmcv/tests/test_ops/test_ms_deformable_attn.py:53-134
(N,M,D=1,2,2; Lq,L,P=2,2,2; shapes [(6,4),(3,2)]; torch.manual_seed(3); value=rand*0.01; ...).
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/VisionLLMv2"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def ast_extract(path, names, glb):
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            ast.fix_missing_locations(mod)
            exec(compile(mod, path, "exec"), glb)
            found[node.name] = glb[node.name]
    missing = set(names) - set(found)
    assert not missing, missing
    return found


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


# --------------------------------------------------------------------------------------------------
def gen_msda():
    glb = {"torch": torch, "F": F}
    fn = ast_extract(f"{REF}/mmcv/mmcv/ops/multi_scale_deform_attn.py",
                     ["multi_scale_deformable_attn_pytorch"], glb)["multi_scale_deformable_attn_pytorch"]

    def case(name, N, M, D, Lq, shapes, P, seed, mode="kat", dtype=torch.float32):
        shapes_t = torch.as_tensor(shapes, dtype=torch.long)
        L = len(shapes)
        S = int(sum(h * w for h, w in shapes))
        torch.manual_seed(seed)
        if mode == "kat":  # exactly the reference test's recipe
            value = torch.rand(N, S, M, D) * 0.01
            loc = torch.rand(N, Lq, M, L, P, 2)
            w = torch.rand(N, Lq, M, L, P) + 1e-5
            w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        else:  # stress: out-of-range / border locations, signed values
            value = torch.randn(N, S, M, D)
            loc = torch.rand(N, Lq, M, L, P, 2) * 1.3 - 0.15
            # force a few exact-border / exact-pixel-centre cases
            flat = loc.view(-1)
            flat[0::97] = 0.0
            flat[1::101] = 1.0
            flat[2::103] = 0.5
            w = torch.softmax(torch.randn(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        out32 = fn(value, shapes_t, loc, w).detach()
        out64 = fn(value.double(), shapes_t, loc.double(), w.double()).detach()
        lsi = torch.cat((shapes_t.new_zeros((1,)), shapes_t.prod(1).cumsum(0)[:-1]))
        np.savez_compressed(os.path.join(OUT, f"msda_{name}.npz"),
                            value=value.numpy(), shapes=shapes_t.numpy(), lsi=lsi.numpy(),
                            loc=loc.numpy(), attw=w.numpy(),
                            out_f32=out32.numpy(), out_f64=out64.numpy(),
                            torch_version=np.array(torch.__version__))
        print(f"msda_{name}: out shape {tuple(out32.shape)}  f64[0,0,:4]={out64[0, 0, :4].tolist()}")

    case("kat_seed3", 1, 2, 2, 2, [(6, 4), (3, 2)], 2, 3, "kat")
    case("stress_small", 2, 4, 8, 37, [(7, 5), (4, 3), (2, 2)], 3, 11, "stress")
    case("stress_d32", 2, 8, 32, 50, [(12, 10), (6, 5), (3, 3), (2, 1)], 4, 12, "stress")
    case("odd_channels", 1, 3, 5, 19, [(5, 6), (3, 3)], 2, 13, "stress")


def ast_extract_class(path, name, glb):
    """exec ONE class definition of a reference file (decorators of the class and of its methods dropped)."""
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == name:
            node.decorator_list = []
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef):
                    sub.decorator_list = [d for d in sub.decorator_list
                                          if isinstance(d, ast.Name) and d.id in ("staticmethod", "classmethod", "property")]
            mod = ast.Module(body=[node], type_ignores=[])
            ast.fix_missing_locations(mod)
            exec(compile(mod, path, "exec"), glb)
            return glb[name]
    raise KeyError(name)


def gen_msda_layer():
    """The deformable-attention LAYER, run from the reference's own module classes on CPU (fp32 and fp64):

      * UniPose ``MSDeformAttn`` (visionllmv2/model/unipose/ops/modules/ms_deform_attn.py:33-145), imported from its file
        with a stub ``MultiScaleDeformableAttention`` extension module whose forward is the reference's own
        ``ms_deform_attn_core_pytorch`` (unipose/ops/functions/ms_deform_attn_func.py:41-61): 2-d reference points,
        4-d reference points, 4-d with ``use_4D_normalizer``;
      * mmcv ``MultiScaleDeformableAttention`` (mmcv/mmcv/ops/multi_scale_deform_attn.py:162-367), AST-extracted class
        (``import mmcv`` needs the compiled ``_ext``); on CPU tensors its forward takes its own torch path (:353-359);
      * ``GroundingDinoMultiscaleDeformableAttention`` (visionllmv2/model/grounding_dino/
        modeling_ov_grounding_dino_mask_dn.py:645-784), AST-extracted class, ``disable_custom_kernels=True``.
    Inputs, parameters and outputs are recorded; tests pin oracle.msda.layer_forward and the three module mirrors."""
    import math
    import warnings
    from typing import Optional
    from torch import Tensor

    # ---- UniPose module, imported from its file under a stub package ----
    func_path = f"{REF}/visionllmv2/model/unipose/ops/functions/ms_deform_attn_func.py"
    mod_path = f"{REF}/visionllmv2/model/unipose/ops/modules/ms_deform_attn.py"
    ext = types.ModuleType("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = ext
    for pkg in ("refops", "refops.functions", "refops.modules"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    spec = importlib.util.spec_from_file_location("refops.functions.ms_deform_attn_func", func_path)
    fmod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = fmod
    spec.loader.exec_module(fmod)
    ext.ms_deform_attn_forward = lambda value, shapes, lsi, loc, w, step: fmod.ms_deform_attn_core_pytorch(value, shapes, loc, w)
    sys.modules["refops.functions"].MSDeformAttnFunction = fmod.MSDeformAttnFunction
    spec = importlib.util.spec_from_file_location("refops.modules.ms_deform_attn", mod_path)
    mmod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mmod
    spec.loader.exec_module(mmod)
    RefMSDeformAttn = mmod.MSDeformAttn

    # ---- mmcv module (AST) ----
    class BaseModule(nn.Module):            # mmcv.runner.BaseModule: only the init_cfg argument is used here
        def __init__(self, init_cfg=None):
            super().__init__()

    glb = {"torch": torch, "nn": nn, "F": F, "math": math, "warnings": warnings, "Optional": Optional,
           "BaseModule": BaseModule, "mmcv": types.SimpleNamespace(ConfigDict=dict)}

    def xavier_init(module, gain=1, bias=0, distribution="normal"):
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
        if module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        nn.init.constant_(module.weight, val)
        if module.bias is not None:
            nn.init.constant_(module.bias, bias)

    glb.update(xavier_init=xavier_init, constant_init=constant_init)
    mm_path = f"{REF}/mmcv/mmcv/ops/multi_scale_deform_attn.py"
    glb.update(ast_extract(mm_path, ["multi_scale_deformable_attn_pytorch"], glb))
    RefMMCV = ast_extract_class(mm_path, "MultiScaleDeformableAttention", glb)

    # ---- Grounding-DINO module (AST) ----
    gd_path = f"{REF}/visionllmv2/model/grounding_dino/modeling_ov_grounding_dino_mask_dn.py"
    glb2 = {"torch": torch, "nn": nn, "F": F, "math": math, "warnings": warnings, "Optional": Optional, "Tensor": Tensor,
            "GroundingDinoConfig": object}
    glb2.update(ast_extract(gd_path, ["multi_scale_deformable_attention"], glb2))
    RefGD = ast_extract_class(gd_path, "GroundingDinoMultiscaleDeformableAttention", glb2)

    shapes = [(9, 7), (5, 4), (3, 2)]
    L, M, P, C, B, Lq = 3, 4, 4, 64, 2, 23
    S = sum(h * w for h, w in shapes)
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))

    def randomise(mod, names, seed):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for n in names:
                lin = getattr(mod, n)
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * (0.5 if "attention" in n else 0.08))
                lin.bias.add_(torch.randn(lin.bias.shape, generator=g) * 0.05)

    def inputs(seed, ref_dim):
        g = torch.Generator().manual_seed(seed)
        query = torch.randn(B, Lq, C, generator=g)
        src = torch.randn(B, S, C, generator=g)
        ref = torch.rand(B, Lq, L, ref_dim, generator=g)
        if ref_dim == 4:
            ref[..., 2:] = ref[..., 2:] * 0.4 + 0.05
        mask = torch.zeros(B, S, dtype=torch.bool)
        mask[1, -9:] = True
        mask[0, 3:6] = True
        return query, src, ref, mask

    out = {"shapes": ss.numpy(), "lsi": lsi.numpy(), "n_heads": np.array(M), "n_levels": np.array(L), "n_points": np.array(P),
           "torch_version": np.array(torch.__version__)}
    names = ["sampling_offsets", "attention_weights", "value_proj", "output_proj"]
    for tag, ref_dim, use4d in (("ref2", 2, False), ("ref4", 4, False), ("ref4_norm", 4, True)):
        torch.manual_seed(0)
        mod = RefMSDeformAttn(d_model=C, n_levels=L, n_heads=M, n_points=P, use_4D_normalizer=use4d).eval()
        randomise(mod, names, 5)
        query, src, ref, mask = inputs(7 + ref_dim + use4d, ref_dim)
        with torch.no_grad():
            y32 = mod(query, ref, src, ss, lsi, mask)
            y64 = mod.double()(query.double(), ref.double(), src.double(), ss, lsi, mask)
        for k, v in mod.float().state_dict().items():
            out[f"unipose_{tag}.sd.{k}"] = v.numpy()
        out.update({f"unipose_{tag}.query": query.numpy(), f"unipose_{tag}.src": src.numpy(), f"unipose_{tag}.ref": ref.numpy(),
                    f"unipose_{tag}.mask": mask.numpy(), f"unipose_{tag}.out_f32": y32.numpy(), f"unipose_{tag}.out_f64": y64.numpy(),
                    f"unipose_{tag}.use4d": np.array(use4d)})
        print(f"msda_layer unipose_{tag}: out {tuple(y32.shape)} |f32-f64| {(y32.double() - y64).abs().max():.2e}")

    # mmcv: (num_query, bs, C) layout (batch_first=False), identity residual, dropout in eval mode = identity, query_pos
    for tag, ref_dim in (("ref2", 2), ("ref4", 4)):
        torch.manual_seed(0)
        mod = RefMMCV(embed_dims=C, num_heads=M, num_levels=L, num_points=P, dropout=0.1).eval()
        randomise(mod, names, 6)
        query, src, ref, mask = inputs(17 + ref_dim, ref_dim)
        g = torch.Generator().manual_seed(3)
        qpos = torch.randn(B, Lq, C, generator=g) * 0.1
        with torch.no_grad():
            y32 = mod(query.transpose(0, 1), value=src.transpose(0, 1), query_pos=qpos.transpose(0, 1), key_padding_mask=mask,
                      reference_points=ref, spatial_shapes=ss, level_start_index=lsi)
            y64 = mod.double()(query.double().transpose(0, 1), value=src.double().transpose(0, 1),
                               query_pos=qpos.double().transpose(0, 1), key_padding_mask=mask, reference_points=ref.double(),
                               spatial_shapes=ss, level_start_index=lsi)
        for k, v in mod.float().state_dict().items():
            out[f"mmcv_{tag}.sd.{k}"] = v.numpy()
        out.update({f"mmcv_{tag}.query": query.numpy(), f"mmcv_{tag}.src": src.numpy(), f"mmcv_{tag}.ref": ref.numpy(),
                    f"mmcv_{tag}.mask": mask.numpy(), f"mmcv_{tag}.query_pos": qpos.numpy(),
                    f"mmcv_{tag}.out_f32": y32.numpy(), f"mmcv_{tag}.out_f64": y64.numpy()})
        print(f"msda_layer mmcv_{tag}: out {tuple(y32.shape)} (num_query, bs, C)")

    # Grounding-DINO: attention_mask is the INVERSE of a padding mask (:733-735), position embeddings added to the query
    for tag, ref_dim in (("ref2", 2), ("ref4", 4)):
        cfg = types.SimpleNamespace(d_model=C, num_feature_levels=L, disable_custom_kernels=True)
        torch.manual_seed(0)
        mod = RefGD(cfg, num_heads=M, n_points=P).eval()
        randomise(mod, names, 8)
        query, src, ref, mask = inputs(27 + ref_dim, ref_dim)
        g = torch.Generator().manual_seed(4)
        pos = torch.randn(B, Lq, C, generator=g) * 0.1
        with torch.no_grad():
            y32, aw32 = mod(query, attention_mask=~mask, encoder_hidden_states=src, position_embeddings=pos, reference_points=ref,
                            spatial_shapes=ss, level_start_index=lsi)
            y64, _ = mod.double()(query.double(), attention_mask=~mask, encoder_hidden_states=src.double(),
                                  position_embeddings=pos.double(), reference_points=ref.double(), spatial_shapes=ss,
                                  level_start_index=lsi)
        for k, v in mod.float().state_dict().items():
            out[f"gdino_{tag}.sd.{k}"] = v.numpy()
        out.update({f"gdino_{tag}.query": query.numpy(), f"gdino_{tag}.src": src.numpy(), f"gdino_{tag}.ref": ref.numpy(),
                    f"gdino_{tag}.mask": mask.numpy(), f"gdino_{tag}.pos": pos.numpy(), f"gdino_{tag}.out_f32": y32.numpy(),
                    f"gdino_{tag}.out_f64": y64.numpy(), f"gdino_{tag}.attw_f32": aw32.numpy()})
        print(f"msda_layer gdino_{tag}: out {tuple(y32.shape)}")
    np.savez_compressed(os.path.join(OUT, "msda_layer.npz"), **out)


def gen_token_loops():
    """The per-sample token loops of VisionLLMv2Model.forward, executed FROM THE REFERENCE'S OWN STATEMENTS: the method cannot
    be imported (peft / mmcv / mmdet / detectron2 ...), so the statement ranges are cut out of its AST and exec'd with a
    stand-in ``self``:
      * modeling_visionllmv2.py:432-524  [EMB] splice behind the tool tokens (training form: the [EMB] ids are present);
      * modeling_visionllmv2.py:779-791  [EMB] hidden states -> text_query / text_query_masks."""
    path = f"{REF}/visionllmv2/model/modeling_visionllmv2.py"
    tree = ast.parse(open(path).read())
    fwd = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "VisionLLMv2Model":
            fwd = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
    assert fwd is not None

    def targets(st):
        return [t.id for t in getattr(st, "targets", []) if isinstance(t, ast.Name)]

    # (1) splice: from the assignment of emb_ids to the two torch.stack statements, in the method body
    body = fwd.body
    i0 = next(i for i, st in enumerate(body) if isinstance(st, ast.Assign) and "emb_ids" in targets(st))
    i1 = next(i for i, st in enumerate(body) if isinstance(st, ast.Assign) and "inputs_embeds" in targets(st) and i > i0 and
              "stack" in ast.unparse(st))
    splice_mod = ast.Module(body=body[i0:i1 + 1], type_ignores=[])
    ast.fix_missing_locations(splice_mod)
    splice_code = compile(splice_mod, path, "exec")

    # (2) text_query: the `emb_select = ...` assignment and the first statements of the `if emb_select.sum() != 0` block
    holders = []
    for node in ast.walk(fwd):
        if isinstance(node, ast.If):
            for k, st in enumerate(node.body):
                if isinstance(st, ast.Assign) and "emb_select" in targets(st) and "input_ids" in ast.unparse(st):
                    holders.append((st.lineno, node.body, k))
    assert holders
    _, hb, k = min(holders, key=lambda t: t[0])                       # the det / grounding branch (:775-787) comes first
    inner = next(st for st in hb[k + 1:] if isinstance(st, ast.If))
    upto = next(i for i, st in enumerate(inner.body) if isinstance(st, ast.For))
    # (the statement ahead of emb_select unpacks batch_size, seq_len, hidden_size from inputs_embeds.shape)
    tq_mod = ast.Module(body=[hb[k - 1], hb[k]] + inner.body[:upto + 1], type_ignores=[])
    ast.fix_missing_locations(tq_mod)
    tq_code = compile(tq_mod, path, "exec")

    torch.manual_seed(21)
    B, L, C, NE, NG = 3, 48, 64, 4, 2
    ids_base = 100
    tool = dict(det_tool_id=11, seg_tool_id=12, grd_tool_id=13, pose_tool_id=14, gen_tool_id=15, edit_tool_id=16)
    emb_token_id = 50
    input_ids = torch.randint(ids_base, ids_base + 30, (B, L))
    def put(b, pos, tool_id, n):
        input_ids[b, pos] = tool_id
        input_ids[b, pos + 1: pos + 1 + n] = torch.arange(emb_token_id, emb_token_id + n) if n == NE else emb_token_id
    put(0, 3, 11, NE); put(0, 20, 14, NE); put(0, 40, 15, NG)
    put(1, 10, 12, NE); put(1, 30, 13, NE)
    put(2, 5, 16, NG)
    inputs_embeds = torch.randn(B, L, C)
    tables = {n: torch.randn(NE if n in ("det", "pose") else NG, C) for n in ("det", "pose", "gen", "edit")}
    self_ = types.SimpleNamespace(emb_token_id=emb_token_id, num_embs=NE, num_embs_gen=NG, **tool,
                                  **{f"emb_embeddings_{n}": types.SimpleNamespace(weight=t) for n, t in tables.items()})
    env = {"torch": torch, "self": self_, "input_ids": input_ids.clone(), "inputs_embeds": inputs_embeds.clone(),
           "gap_len": NE, "gap_len_gen": NG}
    exec(splice_code, env)
    out_ids, out_emb = env["input_ids"], env["inputs_embeds"]

    hidden = torch.randn(B, L, C)
    # (a det / grounding sample carries only [EMB] blocks of num_embs tokens: drop the 2-token generation blocks; sample 2
    # then has no [EMB] token at all -> an all-zero, fully masked row)
    tq_ids = out_ids.clone()
    for b, pos in ((0, 40), (2, 5)):
        tq_ids[b, pos: pos + 1 + NG] = ids_base
    env2 = {"torch": torch, "self": self_, "input_ids": tq_ids.clone(), "inputs_embeds": out_emb, "hidden_states": hidden}
    exec(tq_code, env2)
    print("token loops: splice changed", int((out_emb != inputs_embeds).any(-1).sum()), "rows; text_query", tuple(env2["text_query"].shape))
    np.savez_compressed(os.path.join(OUT, "token_loops.npz"), input_ids=input_ids.numpy(), inputs_embeds=inputs_embeds.numpy(),
                        emb_token_id=np.array(emb_token_id), num_embs=np.array(NE), num_embs_gen=np.array(NG),
                        tool_ids=np.array([tool[k] for k in ("det_tool_id", "seg_tool_id", "grd_tool_id", "pose_tool_id", "gen_tool_id", "edit_tool_id")]),
                        table_det=tables["det"].numpy(), table_pose=tables["pose"].numpy(), table_gen=tables["gen"].numpy(),
                        table_edit=tables["edit"].numpy(), out_ids=out_ids.numpy(), out_embeds=out_emb.numpy(),
                        hidden_states=hidden.numpy(), tq_input_ids=tq_ids.numpy(), text_query=env2["text_query"].numpy(),
                        text_query_masks=env2["text_query_masks"].numpy())


def gen_region_branch():
    """The region branch of VisionLLMv2Model.forward (modeling_visionllmv2.py:609-715), executed FROM THE REFERENCE'S OWN
    STATEMENT: the `if self.use_region_encoder:` block is cut out of the method's AST and exec'd with a stand-in ``self``
    whose ``region_encoder`` records its arguments (all_images, all_regions, all_image_features) and returns seeded features.
    Three input conventions: 'anyres' (list of tile stacks, the last tile is the global image), mmic data (``num_splits``:
    several images per sample, each with its own tile stack; one region per image) and 'pad' (one tensor)."""
    import itertools
    path = f"{REF}/visionllmv2/model/modeling_visionllmv2.py"
    tree = ast.parse(open(path).read())
    fwd = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "VisionLLMv2Model":
            fwd = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
    blocks = [n for n in ast.walk(fwd) if isinstance(n, ast.If) and ast.unparse(n.test) == "self.use_region_encoder"]
    assert len(blocks) == 1 and 605 <= blocks[0].lineno <= 612, [b.lineno for b in blocks]
    mod = ast.Module(body=[blocks[0]], type_ignores=[])
    ast.fix_missing_locations(mod)
    code = compile(mod, path, "exec")

    C, T, reg_token_id = 32, 9, 7          # hidden size, patches per tile, <region> id
    out = {"reg_token_id": np.array(reg_token_id)}

    def run(tag, images, regions, num_splits, split_sizes, n_tiles, seed):
        torch.manual_seed(seed)
        hs = [torch.randn(n_tiles, 1 + T, C) for _ in range(5)]            # encoder hidden states (only the last 3 are read)
        n_all = sum(len(r) for r in regions)
        B, L = len(regions), 24
        input_ids = torch.randint(100, 130, (B, L))
        for b, r in enumerate(regions):                                    # one <region> slot per region, scattered
            pos = torch.randperm(L)[: len(r)]
            input_ids[b, pos] = reg_token_id
        inputs_embeds = torch.randn(B, L, C)
        feats = torch.randn(n_all, C)
        seen = {}

        def region_encoder(all_images, all_regions, all_image_features):
            seen["all_images"], seen["all_regions"] = all_images.clone(), all_regions.clone()
            seen["all_image_features"] = [f.clone() for f in all_image_features]
            return feats

        self_ = types.SimpleNamespace(use_region_encoder=True, region_encoder=region_encoder, reg_token_id=reg_token_id)
        env = {"torch": torch, "itertools": itertools, "self": self_, "regions": regions, "images": images, "num_splits": num_splits,
               "image_forward_outs": types.SimpleNamespace(hidden_states=tuple(hs)), "split_sizes": split_sizes,
               "input_ids": input_ids, "inputs_embeds": inputs_embeds.clone(), "B": B, "L": L, "C": C}
        exec(code, env)
        print(f"region branch [{tag}]: all_images {tuple(seen['all_images'].shape)} features {tuple(seen['all_image_features'][0].shape)}"
              f" x{len(seen['all_image_features'])}; {n_all} <region> slots")
        out.update({f"{tag}_hidden_states": torch.stack(hs).numpy(), f"{tag}_input_ids": input_ids.numpy(),
                    f"{tag}_inputs_embeds": inputs_embeds.numpy(), f"{tag}_region_features": feats.numpy(),
                    f"{tag}_num_regions": np.array([len(r) for r in regions]),
                    f"{tag}_all_images": seen["all_images"].numpy(), f"{tag}_all_regions": seen["all_regions"].numpy(),
                    f"{tag}_all_image_features": torch.stack(seen["all_image_features"]).numpy(),
                    f"{tag}_out_embeds": env["inputs_embeds"].numpy()})
        if split_sizes is not None:
            out[f"{tag}_split_sizes"] = np.array(split_sizes)
        if isinstance(images, list):
            out[f"{tag}_images"] = torch.cat(images, 0).numpy()
        else:
            out[f"{tag}_images"] = images.numpy()

    H = W = 6
    # 'anyres': 3 samples with 3 / 1 / 5 tiles and 2 / 1 / 3 regions
    torch.manual_seed(31)
    split_sizes = [3, 1, 5]
    images = [torch.randn(n, 3, H, W) for n in split_sizes]
    regions = [torch.rand(n, H, W) > 0.5 for n in (2, 1, 3)]
    run("anyres", images, [r.float() for r in regions], None, split_sizes, sum(split_sizes), 32)
    # mmic: sample 0 holds 2 images with 2 + 3 tiles, sample 1 holds 3 images with 1 + 2 + 2 tiles; one region per image,
    # the last image of sample 1 has none (the reference then drops it: [:num_regions[i]])
    num_splits = [[2, 3], [1, 2, 2]]
    split_sizes = [5, 5]
    images = [torch.randn(n, 3, H, W) for n in split_sizes]
    regions = [torch.rand(2, H, W), torch.rand(2, H, W)]
    run("mmic", images, regions, num_splits, split_sizes, sum(split_sizes), 33)
    out["mmic_num_splits_flat"] = np.array([x for ns in num_splits for x in ns])
    out["mmic_num_splits_len"] = np.array([len(ns) for ns in num_splits])
    # 'pad': one tensor of 2 images, 1 / 2 regions
    images = torch.randn(2, 3, H, W)
    regions = [torch.rand(1, H, W), torch.rand(2, H, W)]
    run("pad", images, regions, None, None, 2, 34)
    np.savez_compressed(os.path.join(OUT, "region_branch.npz"), **out)


def gen_dcnv3():
    """DCNv3 forward: the reference's pure-PyTorch twin on the inputs of its own test (ops_dcnv3/test.py:19-66, seed 3:
    N=2, 8x8, M=4, D=16, 3x3, offset_scale 2, pad 1) plus strided / dilated / non-square cases."""
    glb = {"torch": torch, "F": F}
    fns = ast_extract(f"{REF}/visionllmv2/model/ops_dcnv3/functions/dcnv3_func.py",
                      ["_get_reference_points", "_generate_dilation_grids", "dcnv3_core_pytorch"], glb)
    glb.update(fns)
    core = fns["dcnv3_core_pytorch"]

    def case(name, N, H, W, M, D, kh, kw, stride, pad_h, pad_w, dil, offset_scale, seed, off_mag=10.0):
        torch.manual_seed(seed)
        Ho = (H + 2 * pad_h - (dil * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad_w - (dil * (kw - 1) + 1)) // stride + 1
        P = kh * kw
        inp = torch.rand(N, H, W, M * D) * 0.01
        offset = torch.rand(N, Ho, Wo, M * P * 2) * off_mag
        mask = torch.rand(N, Ho, Wo, M, P) + 1e-5
        mask /= mask.sum(-1, keepdim=True)
        mask = mask.reshape(N, Ho, Wo, M * P)
        out64 = core(inp.double(), offset.double(), mask.double(), kh, kw, stride, stride, pad_h, pad_w, dil, dil, M, D,
                     offset_scale).detach()
        out32 = core(inp, offset, mask, kh, kw, stride, stride, pad_h, pad_w, dil, dil, M, D, offset_scale).detach()
        # backward (round 4): autograd through the reference's twin, as its own test does (ops_dcnv3/test.py:94-160), with a random
        # grad_output (its own generator: the forward tensors above keep their values) instead of ones
        gen = torch.Generator().manual_seed(1000 + seed)
        grad_out = torch.randn(out64.shape, generator=gen, dtype=torch.float64)
        grads = {}
        for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
            a, b, c = (t.detach().to(dt).requires_grad_(True) for t in (inp, offset, mask))
            core(a, b, c, kh, kw, stride, stride, pad_h, pad_w, dil, dil, M, D, offset_scale).backward(grad_out.to(dt))
            grads[f"grad_input_{tag}"], grads[f"grad_offset_{tag}"], grads[f"grad_mask_{tag}"] = a.grad.numpy(), b.grad.numpy(), c.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"dcnv3_{name}.npz"), input=inp.numpy(), offset=offset.numpy(),
                            mask=mask.numpy(), out_f32=out32.numpy(), out_f64=out64.numpy(),
                            params=np.array([kh, kw, stride, stride, pad_h, pad_w, dil, dil, M, D], dtype=np.int64),
                            offset_scale=np.array(offset_scale), torch_version=np.array(torch.__version__),
                            grad_out=grad_out.numpy(), **grads)
        print(f"dcnv3_{name}: out {tuple(out64.shape)} f64[0,0,0,:3]={out64[0, 0, 0, :3].tolist()} |grad_offset| max {np.abs(grads['grad_offset_f64']).max():.3g}")

    case("kat_seed3", 2, 8, 8, 4, 16, 3, 3, 1, 1, 1, 1, 2.0, 3)
    case("stride2_dil2", 2, 11, 9, 2, 8, 3, 3, 2, 2, 2, 2, 1.0, 21, off_mag=3.0)
    # (the twin pads W by pad_h and H by pad_w, dcnv3_func.py:129-131, so it only works for pad_h == pad_w)
    case("k5x3_odd_channels", 1, 7, 10, 3, 5, 5, 3, 1, 1, 1, 1, 0.5, 22, off_mag=4.0)
    # the reference's backward channel list (test.py:257-260: 1, 16, 30, 32, 64, 71, 1025) at its test geometry (N 2, 8x8, M 2);
    # 16 / 32 / 64 are covered by the cases above and by the oracle on the GPU side, 1025 by the oracle only (fixture size)
    case("bwd_c1", 2, 8, 8, 2, 1, 3, 3, 1, 1, 1, 1, 2.0, 31)
    case("bwd_c30", 2, 8, 8, 2, 30, 3, 3, 1, 1, 1, 1, 2.0, 32)
    case("bwd_c71", 2, 8, 8, 2, 71, 3, 3, 1, 1, 1, 1, 2.0, 33)


def gen_dcnv3_half():
    """DCNv3 in half precision (round 5; the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF with opmath_t = float,
    ops_dcnv3/src/cuda/dcnv3_cuda.cu:69, 147: half operands, fp32 arithmetic, the result rounded to half).  F.grid_sample has no CPU
    half kernel, so the reference's twin runs in fp32 ON THE HALF-ROUNDED OPERANDS and its result / its autograd gradients are rounded
    to half -- the same quantity up to fp32 rounding before the final rounding.  -> tests/golden/dcnv3_half.npz (three cases)."""
    glb = {"torch": torch, "F": F}
    fns = ast_extract(f"{REF}/visionllmv2/model/ops_dcnv3/functions/dcnv3_func.py",
                      ["_get_reference_points", "_generate_dilation_grids", "dcnv3_core_pytorch"], glb)
    glb.update(fns)
    core = fns["dcnv3_core_pytorch"]
    rec = {}
    for tag, (N, H, W, M, D, k, stride, pad, dil, scale, seed) in {"c16": (2, 8, 8, 4, 16, 3, 1, 1, 1, 2.0, 41),      # vector path
                                                                   "c32": (2, 12, 10, 2, 32, 3, 1, 1, 1, 1.0, 42),    # windowed backward
                                                                   "c5": (1, 7, 10, 3, 5, 3, 2, 1, 1, 0.5, 43)}.items():   # scalar path, stride 2
        torch.manual_seed(seed)
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        P = k * k
        inp = torch.randn(N, H, W, M * D).half()
        offset = (torch.randn(N, Ho, Wo, M * P * 2) * 2.0).half()
        # keep every sampling location away from integer coordinates: there floor() (the kernels) and grid_sample's own convention may
        # pick different cells, and grad_offset is discontinuous across a cell border -- half-rounded offsets hit such points exactly
        c0 = (dil * (k - 1)) >> 1
        for _ in range(4):
            o5 = offset.float().reshape(N, Ho, Wo, M, P, 2)
            pi = torch.arange(k).repeat_interleave(k).float() * dil          # kernel_w outer
            pj = torch.arange(k).repeat(k).float() * dil                     # kernel_h inner
            lw = c0 * (1 - scale) + (pi[None, None, None, None, :] + o5[..., 0]) * scale
            lh = c0 * (1 - scale) + (pj[None, None, None, None, :] + o5[..., 1]) * scale
            near = torch.stack([(lw - lw.round()).abs() < 0.03, (lh - lh.round()).abs() < 0.03], -1)
            if not bool(near.any()):
                break
            offset = (o5 + near.float() * 0.13).reshape(N, Ho, Wo, M * P * 2).half()
        mask = torch.rand(N, Ho, Wo, M, P) + 1e-5
        mask = (mask / mask.sum(-1, keepdim=True)).reshape(N, Ho, Wo, M * P).half()
        grad_out = torch.randn(N, Ho, Wo, M * D).half()
        a, b, c = (t.float().requires_grad_(True) for t in (inp, offset, mask))
        out = core(a, b, c, k, k, stride, stride, pad, pad, dil, dil, M, D, scale)
        out.backward(grad_out.float())
        rec.update({f"{tag}.input": inp.numpy(), f"{tag}.offset": offset.numpy(), f"{tag}.mask": mask.numpy(), f"{tag}.grad_out": grad_out.numpy(),
                    f"{tag}.out": out.detach().half().numpy(), f"{tag}.grad_input": a.grad.half().numpy(), f"{tag}.grad_offset": b.grad.half().numpy(),
                    f"{tag}.grad_mask": c.grad.half().numpy(), f"{tag}.out_f32": out.detach().numpy(),
                    f"{tag}.params": np.array([k, k, stride, stride, pad, pad, dil, dil, M, D], dtype=np.int64), f"{tag}.offset_scale": np.array(scale)})
        print(f"dcnv3_half {tag}: out {tuple(out.shape)} |out| max {float(out.abs().max()):.3g} |grad_offset| max {float(b.grad.abs().max()):.3g}")
    np.savez_compressed(os.path.join(OUT, "dcnv3_half.npz"), torch_version=np.array(torch.__version__), **rec)


def gen_point_sample():
    """Region-encoder point sampling: the reference's own ``point_sample`` (region_encoder.py:24-47) and the masked-mean
    pooling lines (:135-140) on seeded inputs, incl. points on / beyond the border and a region without points."""
    glb = {"torch": torch, "F": F}
    fn = ast_extract(f"{REF}/visionllmv2/model/region_encoder.py", ["point_sample"], glb)["point_sample"]
    torch.manual_seed(5)
    N, C, H, W, P = 3, 6, 7, 5, 40
    x = torch.randn(N, C, H, W)
    pts = torch.rand(N, P, 2) * 1.2 - 0.1
    pts.view(-1)[0::31] = 0.0
    pts.view(-1)[1::37] = 1.0
    valid = torch.rand(N, P) > 0.3
    valid[2] = False
    sampled = fn(x, pts, align_corners=False)
    feats = sampled.permute(0, 2, 1) * valid.unsqueeze(-1)
    pooled = (feats.sum(1) / valid.sum(1).unsqueeze(-1)).nan_to_num()
    np.savez_compressed(os.path.join(OUT, "point_sample.npz"), input=x.numpy(), coords=pts.numpy(), valid=valid.numpy(),
                        sampled=sampled.numpy(), pooled=pooled.numpy(), torch_version=np.array(torch.__version__))
    print("point_sample:", tuple(sampled.shape), pooled[0, :3].tolist())


# --------------------------------------------------------------------------------------------------
def load_intern_vit():
    import transformers.activations, transformers.modeling_outputs, transformers.modeling_utils  # noqa: F401 (before the stub)
    import transformers.configuration_utils  # noqa: F401
    timm = types.ModuleType("timm"); models = types.ModuleType("timm.models"); layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):  # identity stub (eval-mode semantics of timm DropPath)
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x
    layers.DropPath = DropPath
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    pkg = types.ModuleType("refvit"); pkg.__path__ = [f"{REF}/visionllmv2/model/internvit"]
    sys.modules["refvit"] = pkg
    mods = {}
    for name in ("configuration_intern_vit", "modeling_intern_vit"):
        spec = importlib.util.spec_from_file_location(f"refvit.{name}", f"{REF}/visionllmv2/model/internvit/{name}.py")
        m = importlib.util.module_from_spec(spec); sys.modules[f"refvit.{name}"] = m
        spec.loader.exec_module(m); mods[name] = m
    return mods["configuration_intern_vit"].InternVisionConfig, mods["modeling_intern_vit"].InternVisionModel


def gen_intern_vit():
    Cfg, Model = load_intern_vit()

    def case(name, seed, **kw):
        cfg = Cfg(use_flash_attn=False, **kw)
        torch.manual_seed(seed)
        model = Model(cfg).eval().float()
        with torch.no_grad():  # make every parameter non-trivial so a swapped/missing one is caught
            for n_, p in model.named_parameters():
                if n_.endswith("norm1.weight") or n_.endswith("norm2.weight") or "q_norm" in n_ or "k_norm" in n_:
                    p.copy_(1.0 + 0.2 * torch.randn_like(p))
                elif n_.endswith("ls1") or n_.endswith("ls2"):
                    p.copy_(0.1 + 0.05 * torch.randn_like(p))
                elif n_.endswith(".bias"):
                    p.copy_(0.1 * torch.randn_like(p))
                elif "class_embedding" in n_ or "position_embedding" in n_:
                    p.copy_(0.5 * torch.randn_like(p))
        x = torch.randn(2, 3, cfg.image_size, cfg.image_size)
        with torch.no_grad():
            out = model(x, output_hidden_states=True, return_dict=True)
        hs = torch.stack(out.hidden_states, 0)
        cfgd = {k: getattr(cfg, k) for k in ("hidden_size", "num_attention_heads", "num_hidden_layers", "patch_size",
                                             "image_size", "intermediate_size", "layer_norm_eps",
                                             "qk_normalization", "qkv_bias", "hidden_act")}
        np.savez_compressed(os.path.join(OUT, f"internvit_{name}.npz"), pixel_values=x.numpy(),
                            hidden_states=hs.numpy(), last_hidden_state=out.last_hidden_state.numpy(),
                            cfg=np.array(repr(cfgd)), **{"sd." + k: v for k, v in np_sd(model.state_dict()).items()})
        print(f"internvit_{name}: hs {tuple(hs.shape)} |hs[-1]| max {hs[-1].abs().max():.3f}")

    case("tiny_qknorm", 0, hidden_size=64, num_attention_heads=2, intermediate_size=128, num_hidden_layers=3,
         image_size=28, patch_size=14, qk_normalization=True, qkv_bias=False)
    case("tiny_bias", 1, hidden_size=64, num_attention_heads=4, intermediate_size=96, num_hidden_layers=2,
         image_size=42, patch_size=14, qk_normalization=False, qkv_bias=True)
    # GPU-kernel-shaped case: d=64 heads, S=17 (tails everywhere), K multiple of 64
    case("small_d64", 2, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
         image_size=56, patch_size=14, qk_normalization=True, qkv_bias=False)


def gen_clip():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel

    def case(name, seed, **kw):
        cfg = CLIPVisionConfig(**kw)
        try:
            cfg._attn_implementation = "eager"
        except Exception:
            pass
        torch.manual_seed(seed)
        model = CLIPVisionModel(cfg).eval().float()
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if "layer_norm" in n_ or "layrnorm" in n_:
                    p.copy_((1.0 if n_.endswith("weight") else 0.0) + 0.2 * torch.randn_like(p))
                elif n_.endswith(".bias"):
                    p.copy_(0.1 * torch.randn_like(p))
                elif "embedding" in n_:
                    p.copy_(0.5 * torch.randn_like(p))
                else:
                    p.copy_(0.08 * torch.randn_like(p))
        x = torch.randn(2, 3, cfg.image_size, cfg.image_size)
        with torch.no_grad():
            out = model(pixel_values=x, output_hidden_states=True, return_dict=True)
        hs = torch.stack(out.hidden_states, 0)
        cfgd = {k: getattr(cfg, k) for k in ("hidden_size", "num_attention_heads", "num_hidden_layers", "patch_size",
                                             "image_size", "intermediate_size", "layer_norm_eps", "hidden_act")}
        np.savez_compressed(os.path.join(OUT, f"clip_{name}.npz"), pixel_values=x.numpy(), hidden_states=hs.numpy(),
                            cfg=np.array(repr(cfgd)), transformers_version=np.array(transformers.__version__),
                            **{"sd." + k: v for k, v in np_sd(model.state_dict()).items()})
        print(f"clip_{name}: hs {tuple(hs.shape)} keys e.g. {list(model.state_dict())[:3]}")

    case("tiny", 0, hidden_size=64, num_attention_heads=2, intermediate_size=128, num_hidden_layers=3,
         image_size=28, patch_size=14, hidden_act="quick_gelu")
    case("small_d64", 1, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
         image_size=56, patch_size=14, hidden_act="quick_gelu")


def gen_bridge():
    glb = {"torch": torch, "int": int}
    ps = ast_extract(f"{REF}/visionllmv2/model/modeling_visionllmv2.py", ["pixel_shuffle"], glb)["pixel_shuffle"]
    torch.manual_seed(5)
    x = torch.randn(3, 8, 8, 12)
    y = ps(None, x, scale_factor=0.5)
    # bridges constructed exactly as modeling_visionllmv2.py:162-182 does (torch's own modules)
    feats = torch.randn(2, 16, 48)
    torch.manual_seed(6)
    lin = nn.Linear(48, 40)
    mlp2 = nn.Sequential(nn.Linear(48, 40), nn.GELU(), nn.Linear(40, 40))
    ivl = nn.Sequential(nn.LayerNorm(48), nn.Linear(48, 40), nn.GELU(), nn.Linear(40, 40))
    with torch.no_grad():
        ivl[0].weight.copy_(1 + 0.2 * torch.randn(48)); ivl[0].bias.copy_(0.1 * torch.randn(48))
        outs = {"linear": lin(feats), "mlp2x_gelu": mlp2(feats), "internvl_mlp": ivl(feats)}
    save = {"ps_in": x.numpy(), "ps_out": y.numpy(), "feats": feats.numpy()}
    for kind, mod in (("linear", lin), ("mlp2x_gelu", mlp2), ("internvl_mlp", ivl)):
        save[f"out.{kind}"] = outs[kind].numpy()
        for k, v in np_sd(mod.state_dict()).items():
            save[f"sd.{kind}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "bridge.npz"), **save)
    print("bridge: pixel_shuffle", tuple(x.shape), "->", tuple(y.shape))


def gen_tiling():
    glb = {}
    fns = ast_extract(f"{REF}/visionllmv2/mm_utils.py", ["find_closest_aspect_ratio", "dynamic_preprocess"], glb)

    class FakeImg:
        def __init__(self, w, h):
            self.size = (w, h)

        def resize(self, wh):
            return FakeImg(*wh)

        def crop(self, box):
            return FakeImg(box[2] - box[0], box[3] - box[1])

    rows = []
    for (w, h) in [(1336, 1336), (640, 480), (480, 640), (1920, 1080), (300, 1200), (336, 336), (1000, 333), (800, 800)]:
        for (isz, mx) in [(336, 4), (448, 6)]:
            tiles = fns["dynamic_preprocess"](FakeImg(w, h), min_num=1, max_num=mx, image_size=isz, use_thumbnail=True)
            rows.append((w, h, isz, mx, len(tiles)))
    np.savez_compressed(os.path.join(OUT, "tiling.npz"), rows=np.array(rows, dtype=np.int64))
    print("tiling:", rows[:2])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_msda()
    gen_msda_layer()
    gen_token_loops()
    gen_region_branch()
    gen_dcnv3()
    gen_dcnv3_half()
    gen_point_sample()
    gen_intern_vit()
    gen_clip()
    gen_bridge()
    gen_tiling()
