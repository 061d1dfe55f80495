"""ORACLE -- test infrastructure only (never imported by the product).

Plain-torch restatement of the per-sample token loops around the LLM in VisionLLMv2Model.forward
(VisionLLMv2/visionllmv2/model/modeling_visionllmv2.py), pinned by tests/test_oracle_tokens.py to fixtures produced by
executing the reference's own statements (oracle/gen_golden.py::gen_token_loops):
  * emb_splice        :432-524   [EMB] query tables written behind the tool tokens (training form, [EMB] ids present)
  * text_query        :775-787   [EMB] hidden states gathered per sample into the det head's text_query + mask
  * region_features / region_images :626-676   per region, its image's global tile ('anyres', mmic num_splits, 'pad'), last three levels
  * region_splice     :688-695   <region> slots take the region encoder's output"""
import torch


def emb_splice(input_ids, inputs_embeds, tool_tables, emb_token_id, num_embs, num_embs_gen, gen_tools=()):
    """tool_tables: ordered {tool_id: table}; the reference walks det/seg/grd, pose, gen, edit in this order per sample."""
    ids_out, emb_out = [], []
    for cur_ids, cur_emb in zip(input_ids, inputs_embeds):
        new_ids, new_emb = cur_ids, cur_emb
        for tool_id, table in tool_tables.items():
            gen = tool_id in gen_tools
            gap = num_embs_gen if gen else num_embs
            emb_ids = (emb_token_id * torch.ones(num_embs_gen, dtype=torch.long) if gen
                       else torch.arange(emb_token_id, emb_token_id + num_embs, dtype=torch.long))
            for start in torch.where(cur_ids == tool_id)[0]:
                new_ids = torch.cat([new_ids[: start + 1], emb_ids, new_ids[start + gap + 1:]], 0)
                new_emb = torch.cat([new_emb[: start + 1], table.to(new_emb.dtype), new_emb[start + gap + 1:]], 0)
        ids_out.append(new_ids)
        emb_out.append(new_emb)
    return torch.stack(ids_out, 0), torch.stack(emb_out, 0)


def text_query(hidden_states, input_ids, emb_token_id, num_embs):
    B, L, C = hidden_states.shape
    sel = (input_ids >= emb_token_id) & (input_ids <= emb_token_id + num_embs - 1)
    if sel.sum() == 0:
        return None, None
    num_patches = sel.sum(-1) // num_embs
    mx = int(num_patches.max())
    tq = torch.zeros((B, mx, num_embs, C), dtype=hidden_states.dtype)
    masks = torch.zeros(B, mx, dtype=torch.bool)
    for b in range(B):
        if num_patches[b] != 0:
            tq[b, : num_patches[b]] = hidden_states[b, sel[b], :].reshape(-1, num_embs, C)
            masks[b, : num_patches[b]] = 1
    return tq, masks


def region_features(hidden_states, split_sizes, num_regions, levels=(-3, -2, -1), num_splits=None):
    """all_image_features (:644-676).  split_sizes None: 'pad' (one image per sample); num_splits: mmic data."""
    import itertools
    outs = []
    for lv in levels:
        hs = hidden_states[lv]
        if num_splits is not None:                                                   # :646-662
            per_sample = torch.split(hs, split_sizes, dim=0)
            rows = []
            for i, (x, ns) in enumerate(zip(per_sample, num_splits)):
                last = torch.as_tensor([c - 1 for c in itertools.accumulate(ns)], dtype=torch.long)
                rows.append(x[last, 1:][: num_regions[i]])
            outs.append(torch.cat(rows, 0))
            continue
        if split_sizes is not None:                                                  # 'anyres' :664-671
            per_sample = torch.split(hs, split_sizes, dim=0)
            glob = torch.stack([x[-1, 1:] for x in per_sample], 0)                   # [bs, img_len, C]
        else:                                                                        # 'pad' :672-673
            glob = hs[:, 1:]
        outs.append(torch.cat([glob[i][None].repeat_interleave(num_regions[i], dim=0) for i in range(glob.shape[0])]))
    return outs


def region_images(images, num_regions, num_splits=None):
    """all_images (:626-643)."""
    import itertools
    if num_splits is not None:
        rows = []
        for i, (x, ns) in enumerate(zip(images, num_splits)):
            last = torch.as_tensor([c - 1 for c in itertools.accumulate(ns)], dtype=torch.long)
            rows.append(x[last][: num_regions[i]])
        return torch.cat(rows, 0)
    if isinstance(images, (list, tuple)):
        return torch.cat([images[i][-1][None].repeat_interleave(num_regions[i], dim=0) for i in range(len(images))], 0)
    return torch.cat([images[i][None].repeat_interleave(num_regions[i], dim=0) for i in range(len(images))], 0)


def region_splice(inputs_embeds, input_ids, reg_token_id, region_feats):
    B, L, C = inputs_embeds.shape
    flat = inputs_embeds.reshape(B * L, C)
    mask = (input_ids == reg_token_id).reshape(-1)
    temp = torch.zeros_like(flat)
    temp[mask] = region_feats.to(flat.dtype)
    m = mask.to(flat.dtype).unsqueeze(-1)
    return (flat * (1 - m) + temp * m).reshape(B, L, C)
