"""oracle/tokens.py (the restated per-sample token loops) is pinned to fixtures produced by EXECUTING the reference's own
statements (oracle/gen_golden.py::gen_token_loops, modeling_visionllmv2.py:432-524 and :775-787)."""
import numpy as np
import torch

from conftest import load_golden
from oracle import tokens as T


def _tables(g):
    det, seg, grd, pose, gen, edit = [int(x) for x in g["tool_ids"]]
    t = {k: torch.from_numpy(g["table_" + k]) for k in ("det", "pose", "gen", "edit")}
    return {det: t["det"], seg: t["det"], grd: t["det"], pose: t["pose"], gen: t["gen"], edit: t["edit"]}, (gen, edit)


def test_emb_splice_matches_reference_statements():
    g = load_golden("token_loops.npz")
    tables, gen_tools = _tables(g)
    ids, emb = T.emb_splice(torch.from_numpy(g["input_ids"]), torch.from_numpy(g["inputs_embeds"]), tables, int(g["emb_token_id"]),
                            int(g["num_embs"]), int(g["num_embs_gen"]), gen_tools)
    assert np.array_equal(ids.numpy(), g["out_ids"]) and np.array_equal(emb.numpy(), g["out_embeds"])


def test_text_query_matches_reference_statements():
    g = load_golden("token_loops.npz")
    tq, masks = T.text_query(torch.from_numpy(g["hidden_states"]), torch.from_numpy(g["tq_input_ids"]), int(g["emb_token_id"]),
                             int(g["num_embs"]))
    assert np.array_equal(tq.numpy(), g["text_query"]) and np.array_equal(masks.numpy(), g["text_query_masks"])
