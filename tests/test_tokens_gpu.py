"""GPU parity of the per-sample token loops: product (visionllm_amd/splice.py: index bookkeeping +
the native row mover vllm_copy_rows_bf16) vs fixtures produced by executing the reference's statements, and vs the oracle
restatement on larger random cases.  Row movement is a copy: every comparison is BIT-EXACT."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tokens as T
from visionllm_amd import splice as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tables(g, dev=None, dtype=torch.bfloat16):
    det, seg, grd, pose, gen, edit = [int(x) for x in g["tool_ids"]]
    t = {k: torch.from_numpy(g["table_" + k]).to(dtype) for k in ("det", "pose", "gen", "edit")}
    if dev:
        t = {k: v.to(dev) for k, v in t.items()}
    return {det: t["det"], seg: t["det"], grd: t["det"], pose: t["pose"], gen: t["gen"], edit: t["edit"]}, (gen, edit)


def test_emb_splice_vs_reference_fixture():
    g = load_golden("token_loops.npz")
    emb = torch.from_numpy(g["inputs_embeds"]).to(torch.bfloat16)
    tables_cpu, gen_tools = _tables(g)
    # the fixture is fp32; the same statements on bf16-rounded inputs (exact copies -> bit-exact in bf16 too)
    ids_ref, emb_ref = T.emb_splice(torch.from_numpy(g["input_ids"]), emb, tables_cpu, int(g["emb_token_id"]), int(g["num_embs"]),
                                    int(g["num_embs_gen"]), gen_tools)
    assert np.array_equal(ids_ref.numpy(), g["out_ids"])
    assert torch.equal(emb_ref, torch.from_numpy(g["out_embeds"]).to(torch.bfloat16))
    tables, _ = _tables(g, DEV)
    ids, out = S.splice_emb_tokens(torch.from_numpy(g["input_ids"]).to(DEV), emb.to(DEV).clone(), tables, int(g["emb_token_id"]),
                                   int(g["num_embs"]), int(g["num_embs_gen"]), gen_tools)
    assert torch.equal(ids.cpu(), ids_ref) and torch.equal(out.cpu(), emb_ref)


def test_text_query_vs_reference_fixture_and_random_cases():
    g = load_golden("token_loops.npz")
    hs = torch.from_numpy(g["hidden_states"]).to(torch.bfloat16)
    ids = torch.from_numpy(g["tq_input_ids"])
    tq, masks = S.gather_emb_hidden_states(hs.to(DEV), ids.to(DEV), int(g["emb_token_id"]), int(g["num_embs"]))
    assert torch.equal(tq.cpu(), torch.from_numpy(g["text_query"]).to(torch.bfloat16))
    assert np.array_equal(masks.cpu().numpy(), g["text_query_masks"])
    # larger random case: 0..5 patches per sample at random places
    torch.manual_seed(5)
    B, L, C, NE, E0 = 6, 300, 256, 4, 900
    ids = torch.randint(0, 800, (B, L))
    for b in range(B):
        for k in range(b % 6):
            p = 10 + 40 * k + b
            ids[b, p:p + NE] = torch.arange(E0, E0 + NE)
    hs = torch.randn(B, L, C).to(torch.bfloat16)
    ref, rm = T.text_query(hs, ids, E0, NE)
    tq, masks = S.gather_emb_hidden_states(hs.to(DEV), ids.to(DEV), E0, NE)
    assert torch.equal(tq.cpu(), ref) and torch.equal(masks.cpu(), rm)
    assert S.gather_emb_hidden_states(hs.to(DEV), torch.zeros_like(ids).to(DEV), E0, NE) == (None, None)
    bad = ids.clone()
    bad[1, 0] = E0                                   # whole patches + a stray [EMB] token (the reference's reshape fails)
    with pytest.raises(RuntimeError):
        S.gather_emb_hidden_states(hs.to(DEV), bad.to(DEV), E0, NE)
    lone = ids.clone()
    lone[0, 0] = E0                                  # a sample with ONLY a stray token: num_patches == 0, skipped by the reference (:783-786)
    tq2, m2 = S.gather_emb_hidden_states(hs.to(DEV), lone.to(DEV), E0, NE)
    r2, rm2 = T.text_query(hs, lone, E0, NE)
    assert torch.equal(tq2.cpu(), r2) and torch.equal(m2.cpu(), rm2)


def test_region_feature_gather_and_region_splice_vs_oracle():
    torch.manual_seed(9)
    split_sizes, num_regions = [5, 1, 3], [2, 0, 3]
    n_tiles, S1, C = sum(split_sizes), 1 + 16, 128
    hs = [torch.randn(n_tiles, S1, C).to(torch.bfloat16) for _ in range(4)]
    ref = T.region_features(hs, split_sizes, num_regions)
    out = S.gather_region_image_features([h.to(DEV) for h in hs], split_sizes, num_regions)
    assert len(out) == 3 and all(torch.equal(o.cpu(), r) for o, r in zip(out, ref))
    B, L = 3, 40
    REG = 77
    ids = torch.randint(100, 200, (B, L))
    ids[0, 3] = ids[0, 9] = REG
    ids[2, 1] = ids[2, 2] = ids[2, 30] = REG
    emb = torch.randn(B, L, C).to(torch.bfloat16)
    feats = torch.randn(5, C).to(torch.bfloat16)
    ref = T.region_splice(emb.float(), ids, REG, feats.float()).to(torch.bfloat16)
    out = S.splice_region_tokens(emb.to(DEV).clone(), ids.to(DEV), REG, feats.to(DEV))
    assert torch.equal(out.cpu(), ref)
    with pytest.raises(RuntimeError):
        S.splice_region_tokens(emb.to(DEV).clone(), ids.to(DEV), REG, feats[:4].to(DEV))


def test_emb_splice_random_vs_oracle_and_errors():
    torch.manual_seed(3)
    B, L, C, NE, NG, E0 = 4, 120, 192, 8, 3, 500
    ids = torch.randint(0, 400, (B, L))
    tools = {450: "det", 451: "det", 453: "pose", 454: "gen"}
    for b in range(B):
        for k, tid in enumerate(list(tools)[: 1 + b]):
            p = 5 + 25 * k
            n = NG if tools[tid] == "gen" else NE
            ids[b, p] = tid
            ids[b, p + 1: p + 1 + n] = E0 if tools[tid] == "gen" else torch.arange(E0, E0 + n)
    tabs = {"det": torch.randn(NE, C), "pose": torch.randn(NE, C), "gen": torch.randn(NG, C)}
    tables = {tid: tabs[k].to(torch.bfloat16) for tid, k in tools.items()}
    emb = torch.randn(B, L, C).to(torch.bfloat16)
    ids_ref, emb_ref = T.emb_splice(ids, emb, tables, E0, NE, NG, gen_tools=(454,))
    ids_out, emb_out = S.splice_emb_tokens(ids.to(DEV), emb.to(DEV).clone(), {k: v.to(DEV) for k, v in tables.items()}, E0, NE, NG,
                                           gen_tools=(454,))
    assert torch.equal(ids_out.cpu(), ids_ref) and torch.equal(emb_out.cpu(), emb_ref)
    ids[0, L - 3] = 450                               # a tool token too close to the end of the sequence
    with pytest.raises(RuntimeError):
        S.splice_emb_tokens(ids.to(DEV), emb.to(DEV).clone(), {k: v.to(DEV) for k, v in tables.items()}, E0, NE, NG, gen_tools=(454,))


def test_region_branch_vs_reference_fixture():
    """Region branch (modeling_visionllmv2.py:609-715) against the fixture made by executing the reference's statement:
    image / feature selection for 'anyres', mmic (num_splits) and 'pad' inputs, and the <region> splice -- bit-exact."""
    from test_oracle_tokens import _region_case
    g = load_golden("region_branch.npz")
    for tag in ("anyres", "mmic", "pad"):
        hs, images, split_sizes, num_regions, num_splits = _region_case(g, tag)
        hs_dev = [h.to(torch.bfloat16).to(DEV) for h in hs]
        feats = S.gather_region_image_features(hs_dev, split_sizes, num_regions, num_splits=num_splits)
        want = torch.from_numpy(g[f"{tag}_all_image_features"]).to(torch.bfloat16)
        assert torch.equal(torch.stack(feats).cpu(), want), tag
        imgs = [x.to(DEV) for x in images] if isinstance(images, list) else images.to(DEV)
        assert torch.equal(S.gather_region_images(imgs, num_regions, num_splits).cpu(), torch.from_numpy(g[f"{tag}_all_images"])), tag
        emb = torch.from_numpy(g[f"{tag}_inputs_embeds"]).to(torch.bfloat16).to(DEV)
        rf = torch.from_numpy(g[f"{tag}_region_features"]).to(torch.bfloat16).to(DEV)
        out = S.splice_region_tokens(emb, torch.from_numpy(g[f"{tag}_input_ids"]).to(DEV), int(g["reg_token_id"]), rf)
        # (bf16-rounded inputs: rows are copies, so rounding commutes with the reference's x * (1 - m) + t * m)
        want = torch.from_numpy(g[f"{tag}_out_embeds"]).to(torch.bfloat16)
        got = out.cpu()
        ids = torch.from_numpy(g[f"{tag}_input_ids"])
        slot = ids == int(g["reg_token_id"])
        assert torch.equal(got[~slot], want[~slot]) and torch.equal(got[slot], rf.cpu()), tag


def test_token_loops_refuse_to_cut_gradients():
    """the native row mover has no backward -- inputs that require grad raise instead of losing it."""
    C = 64
    emb = torch.randn(1, 8, C, device=DEV, dtype=torch.bfloat16)
    ids = torch.tensor([[5, 1, 2, 3, 4, 9, 9, 9]], device=DEV)
    table = torch.randn(2, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        S.splice_emb_tokens(ids, emb.clone(), {5: table}, 100, 2)
    with torch.no_grad():
        S.splice_emb_tokens(ids, emb.clone(), {5: table}, 100, 2)
    hs = torch.randn(1, 8, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        S.gather_emb_hidden_states(hs, ids, 1, 2)
    with pytest.raises(RuntimeError, match="forward-only"):
        S.splice_region_tokens(emb.clone(), ids, 9, torch.randn(3, C, device=DEV, dtype=torch.bfloat16, requires_grad=True))
    with pytest.raises(RuntimeError, match="forward-only"):
        S.gather_region_image_features([torch.randn(2, 5, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)] * 3, [2], [1])
    # a sample with fewer stray [EMB] tokens than one patch is skipped (the reference's num_patches == 0), not an error
    ids2 = torch.tensor([[1, 2, 7, 7, 7, 7, 7, 7], [1, 7, 7, 7, 7, 7, 7, 7]], device=DEV)
    hs2 = torch.randn(2, 8, C, device=DEV, dtype=torch.bfloat16)
    tq, masks = S.gather_emb_hidden_states(hs2, ids2, 1, 2)
    assert tq.shape == (2, 1, 2, C) and masks.tolist() == [[True], [False]] and bool((tq[1] == 0).all())
