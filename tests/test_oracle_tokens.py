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


def _region_case(g, tag):
    hs = [torch.from_numpy(x) for x in g[f"{tag}_hidden_states"]]
    num_regions = [int(x) for x in g[f"{tag}_num_regions"]]
    split_sizes = [int(x) for x in g[f"{tag}_split_sizes"]] if f"{tag}_split_sizes" in g else None
    num_splits = None
    if tag == "mmic":
        flat, lens = [int(x) for x in g["mmic_num_splits_flat"]], [int(x) for x in g["mmic_num_splits_len"]]
        num_splits, k = [], 0
        for n in lens:
            num_splits.append(flat[k:k + n])
            k += n
    images = torch.from_numpy(g[f"{tag}_images"])
    if split_sizes is not None:
        images = list(torch.split(images, split_sizes, dim=0))
    return hs, images, split_sizes, num_regions, num_splits


def test_region_branch_matches_reference_statements():
    """modeling_visionllmv2.py:609-715 executed from the reference's own AST (gen_golden.py::gen_region_branch): what the
    region encoder receives (all_images, all_image_features) and the <region> splice, for 'anyres', mmic (num_splits) and
    'pad' inputs."""
    g = load_golden("region_branch.npz")
    for tag in ("anyres", "mmic", "pad"):
        hs, images, split_sizes, num_regions, num_splits = _region_case(g, tag)
        feats = T.region_features(hs, split_sizes, num_regions, num_splits=num_splits)
        assert np.array_equal(torch.stack(feats).numpy(), g[f"{tag}_all_image_features"]), tag
        assert np.array_equal(T.region_images(images, num_regions, num_splits).numpy(), g[f"{tag}_all_images"]), tag
        out = T.region_splice(torch.from_numpy(g[f"{tag}_inputs_embeds"]), torch.from_numpy(g[f"{tag}_input_ids"]),
                              int(g["reg_token_id"]), torch.from_numpy(g[f"{tag}_region_features"]))
        assert np.array_equal(out.numpy(), g[f"{tag}_out_embeds"]), tag


def test_region_tile_index_host_logic():
    """The product's index arithmetic (visionllm_amd/splice.py::region_tile_index; plain Python, no GPU)."""
    from visionllm_amd.splice import region_tile_index as rti
    assert rti([3, 1, 5], [2, 1, 3]) == [2, 2, 3, 8, 8, 8]                       # 'anyres': last tile of each sample
    assert rti([5, 5], [2, 2], [[2, 3], [1, 2, 2]]) == [1, 4, 5, 7]              # mmic: global tile of each image, first n
    assert rti(None, [1, 2], n_images=2) == [0, 1, 1]                            # 'pad'
    assert rti([2, 2, 2, 2], [1, 2]) == [1, 3, 3, 5, 7, 7]                       # num_beams = 2 (:617-621)
    import pytest
    with pytest.raises(RuntimeError):
        rti([3, 0], [1, 1])                                                      # a sample without tiles (reference: x[-1] raises)
    with pytest.raises(RuntimeError):
        rti([5], [3], [[2, 3]])                                                  # more regions than images
    with pytest.raises(RuntimeError):
        rti([5], [1], [[2, 2]])                                                  # num_splits does not add up
