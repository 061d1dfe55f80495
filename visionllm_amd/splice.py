"""Visual-token splice: writes the projector output into the ``<im_patch>`` slots of the LLM input embeddings
(VisionLLMv2/visionllmv2/model/modeling_visionllmv2.py:582-605).  The reference does it with boolean-mask indexing
(``image_features[has_image]``, ``inputs_embeds[selected] = inputs_embeds[selected] * 0.0 + vit_embeds``: two host
synchronisations, two full copies); here the whole splice -- slot list, ``has_image``, the tiles of the samples that have an image,
the token-count rule, the row movement -- is ONE native call (``vllm_splice_visual_tokens_bf16``, round 5; rounds 1-4:
``torch.nonzero`` + a scatter kernel).  The call itself does not synchronise; with ``check=True`` (the default, and what bench.py
times) the wrapper then reads the 4-int status word back, which IS a host synchronisation -- the one the reference's error
behaviour costs.  ``check=False`` skips it and hands the status tensor to the caller instead (``status=`` / ``return_status``).
Limit: B <= 4096 samples per call (the slot scan is one block; the torch implementation of rounds 1-4 had none)."""
import ctypes

import torch

from . import _lib


def splice_visual_tokens(inputs_embeds, input_ids, imp_token_id, image_features, split_sizes=None, check=True, return_status=False):
    """inputs_embeds [B, L, C] (bf16, CUDA, modified in place and returned), input_ids [B, L],
    image_features [n_tiles, T, C] in tile order, split_sizes: tiles per sample ('anyres' list input) or None.

    Mirrors the reference's handling of samples without an image (their tiles are dropped, :585-592) and of a
    token-count mismatch (:597-603): features are repeated when the slots are a whole multiple of them; any other mismatch
    raises, as the reference's second assignment does (and nothing has been written).  ``check=False`` skips reading the status
    word back (the only host synchronisation left) -- for callers that validated the prompt on the host, or capture the step in a
    graph; a mismatch then leaves ``inputs_embeds`` untouched, and ``return_status=True`` returns ``(inputs_embeds, status)`` with
    ``status`` the DEVICE int32 tensor ``[_, n_visual_tokens, mismatch, n_slots]`` so that it can be checked later
    (``splice_status_ok(status)``) without a synchronisation now."""
    B, L, C = inputs_embeds.shape
    if not inputs_embeds.is_cuda or inputs_embeds.dtype != torch.bfloat16 or not inputs_embeds.is_contiguous():
        raise RuntimeError("splice_visual_tokens: inputs_embeds must be a contiguous bf16 CUDA tensor")
    dev = inputs_embeds.device
    ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
    if tuple(ids.shape) != (B, L):
        raise RuntimeError(f"splice_visual_tokens: input_ids {tuple(ids.shape)} does not match inputs_embeds {(B, L, C)}")
    feats = image_features.to(device=dev, dtype=torch.bfloat16)
    if feats.dim() != 3 or feats.shape[-1] != C:
        raise RuntimeError(f"splice_visual_tokens: image_features must be [n_tiles, T, {C}], got {tuple(feats.shape)}")
    feats = feats.contiguous()
    n_tiles, T = int(feats.shape[0]), int(feats.shape[1])
    tps = None
    if split_sizes is not None:
        if len(split_sizes) != B:
            raise RuntimeError("splice_visual_tokens: one split size per sample required")
        tps = (ctypes.c_int32 * B)(*[int(v) for v in split_sizes])   # host array: travels as a kernel argument
    L_ = _lib.lib()
    ws = torch.empty(int(L_.vllm_splice_workspace_ints(B, L, n_tiles)), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L_.vllm_splice_visual_tokens_bf16(_lib.ptr(ids), int(imp_token_id), _lib.ptr(feats), tps, B, L, n_tiles, T, C,
                                                     _lib.ptr(inputs_embeds), _lib.ptr(ws), None, _lib.current_stream(dev)),
                   "vllm_splice_visual_tokens_bf16")
    if check:
        _, n_vit, bad, n_sel = (int(v) for v in ws[:4].cpu())
        if bad:
            raise RuntimeError(f"splice_visual_tokens: shape mismatch: {n_sel} <im_patch> slots cannot take {n_vit} visual tokens")
    if return_status:
        return inputs_embeds, ws[:4]
    return inputs_embeds


def splice_status_ok(status):
    """Deferred form of splice_visual_tokens' check: reads a status tensor returned with ``return_status=True`` (synchronises) and
    raises the reference's shape-mismatch error if that splice wrote nothing."""
    _, n_vit, bad, n_sel = (int(v) for v in status.cpu())
    if bad:
        raise RuntimeError(f"splice_visual_tokens: shape mismatch: {n_sel} <im_patch> slots cannot take {n_vit} visual tokens")
    return True


# ---- the other per-sample token loops around the LLM: index bookkeeping in torch, row movement
# ---- by ONE native kernel (vllm_copy_rows_bf16: dst[dst_idx[i]] = src[src_idx[i]]) ------------------------------------------
def _copy_rows(src, src_idx, dst, dst_idx, n):
    """dst / src: contiguous bf16 CUDA tensors viewed as rows of C; indices: int64 CUDA tensors or None (identity)."""
    C = src.shape[-1]
    if not (src.is_cuda and dst.is_cuda and src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16 and
            src.is_contiguous() and dst.is_contiguous() and dst.shape[-1] == C):
        raise RuntimeError("copy_rows: contiguous bf16 CUDA tensors with equal row length required")
    if n == 0:
        return dst
    si = None if src_idx is None else src_idx.to(torch.int64).contiguous()
    di = None if dst_idx is None else dst_idx.to(torch.int64).contiguous()
    with torch.cuda.device(dst.device):
        _lib.check(_lib.lib().vllm_copy_rows_bf16(_lib.ptr(src), _lib.ptr(si), _lib.ptr(dst), _lib.ptr(di), int(n), C,
                                                  src.numel() // C, dst.numel() // C, _lib.current_stream(dst.device)),
                   "vllm_copy_rows_bf16")
    return dst


def splice_emb_tokens(input_ids, inputs_embeds, tool_tables, emb_token_id, num_embs, num_embs_gen=None, gen_tools=()):
    """[EMB] splice (modeling_visionllmv2.py:425-527), the form in which the [EMB] tokens are already present in
    ``input_ids`` (gap_len = num_embs: prefill of a prompt that carries them; FORWARD ONLY -- raises when an input requires grad): behind every tool token the next ``num_embs`` rows of ``inputs_embeds``
    are REPLACED by that tool's learned query table and the ids by the [EMB] id range.

    ``tool_tables``: ordered ``{tool_token_id: table [num_embs, C]}`` in the reference's order of application (det, seg,
    grd -> emb_embeddings_det; pose -> emb_embeddings_pose; gen -> emb_embeddings_gen; edit -> emb_embeddings_edit); tool ids
    listed in ``gen_tools`` use ``num_embs_gen`` rows and the single id ``emb_token_id`` (:436).  Returns
    (input_ids, inputs_embeds): ids as a new tensor, embeddings modified IN PLACE.  A table that would run past the end of
    the sequence raises (the reference's torch.stack of unequal lengths does)."""
    B, L, C = inputs_embeds.shape
    _no_grad_path("splice_emb_tokens", inputs_embeds, *tool_tables.values())
    ids = input_ids.clone()
    flat = inputs_embeds.view(B * L, C)
    for tool_id, table in tool_tables.items():
        n_rows = num_embs_gen if tool_id in gen_tools else num_embs
        if table.shape[0] != n_rows:
            raise RuntimeError(f"splice_emb_tokens: table of tool {tool_id} has {table.shape[0]} rows, expected {n_rows}")
        pos = torch.nonzero(input_ids == tool_id, as_tuple=False)            # [n, 2] (batch, position)
        if pos.numel() == 0:
            continue
        if int((pos[:, 1] + n_rows).max()) >= L:
            raise RuntimeError("splice_emb_tokens: [EMB] block runs past the end of the sequence")
        offs = torch.arange(1, n_rows + 1, device=pos.device)
        dst = ((pos[:, 0] * L + pos[:, 1])[:, None] + offs[None, :]).reshape(-1)             # flattened target rows
        src = offs.sub(1).repeat(pos.shape[0])                                               # table rows, repeated per tool token
        _copy_rows(table.detach().to(inputs_embeds.dtype).contiguous(), src, flat, dst, dst.numel())
        new_ids = (torch.full((n_rows,), emb_token_id, device=ids.device, dtype=ids.dtype) if tool_id in gen_tools else
                   torch.arange(emb_token_id, emb_token_id + n_rows, device=ids.device, dtype=ids.dtype))
        ids.view(-1)[dst] = new_ids.repeat(pos.shape[0])
    return ids, inputs_embeds


def gather_emb_hidden_states(hidden_states, input_ids, emb_token_id, num_embs):
    """[EMB] hidden states -> the det head's text_query (modeling_visionllmv2.py:775-787): returns
    (text_query [B, max_patches, num_embs, C] zero padded, text_query_masks [B, max_patches] bool), or (None, None) when no
    [EMB] token is present.  The reference loops over the batch; here the k-th selected token of sample b goes to row
    b * max_patches * num_embs + k in one native row copy (one host read of the patch counts, as the reference's .max())."""
    B, L, C = hidden_states.shape
    _no_grad_path("gather_emb_hidden_states", hidden_states)
    sel = (input_ids >= emb_token_id) & (input_ids <= emb_token_id + num_embs - 1)
    counts = sel.sum(-1)
    total = int(counts.sum())
    if total == 0:
        return None, None
    num_patches = counts // num_embs
    max_p = int(num_patches.max())
    out = torch.zeros((B, max_p, num_embs, C), dtype=hidden_states.dtype, device=hidden_states.device)
    masks = torch.arange(max_p, device=hidden_states.device)[None, :] < num_patches[:, None]
    if max_p == 0:
        return out, masks
    rank = sel.cumsum(-1) - 1                                                   # position of a selected token within its sample
    keep = sel & (rank < (num_patches * num_embs)[:, None])                     # (the reference's reshape(-1, num_embs, C) needs whole patches)
    # a sample with 0 < count < num_embs is SKIPPED by the reference (num_patches == 0, :783-786); one with whole patches
    # plus a remainder fails its reshape(-1, num_embs, C)
    if bool(((counts % num_embs != 0) & (num_patches > 0)).any()):
        raise RuntimeError("gather_emb_hidden_states: a sample's [EMB] tokens are not a whole number of patches")
    src = torch.nonzero(keep.reshape(-1), as_tuple=False).reshape(-1)
    b_of = src // L
    dst = b_of * (max_p * num_embs) + rank.reshape(-1)[src]
    _copy_rows(hidden_states.contiguous().view(B * L, C), src, out.view(-1, C), dst, src.numel())
    return out, masks


def _no_grad_path(what, *tensors):
    """The native row mover has no backward: refuse to cut a gradient silently (the reference trains emb_embeddings_*
    through the [EMB] splice, back-propagates the det-head loss through the [EMB] hidden states into the LLM
    (modeling_visionllmv2.py:775-787) and the region features into the region encoder)."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError(f"{what}: forward-only (inference / prefill) -- an input requires grad; run it under "
                           "torch.no_grad() or keep the reference's torch indexing for training")


def region_tile_index(split_sizes, num_regions, num_splits=None, n_images=None):
    """Which tile (row of the concatenated tile batch) feeds each region, as a Python list -- the index arithmetic of the
    reference's three input conventions (modeling_visionllmv2.py:626-676):
      * 'anyres' (``split_sizes``: tiles per sample, ``num_splits`` None): the LAST tile of the sample (the global image),
        once per region of the sample;
      * mmic data (``num_splits``: per sample, tiles per image): the global tile of each image (cumulative sums - 1), the
        first ``num_regions[i]`` of them (one region per image);
      * 'pad' (``split_sizes`` None): image i itself, once per region.
    ``len(split_sizes)`` (or ``n_images``) may be a multiple of ``len(num_regions)``: generate() with num_beams > 1 (:617-621)."""
    num_regions = [int(n) for n in num_regions]
    n_samples = len(split_sizes) if split_sizes is not None else int(n_images)
    if len(num_regions) == 0 or n_samples % len(num_regions) != 0:
        raise RuntimeError(f"region_tile_index: {n_samples} samples vs {len(num_regions)} region lists")
    num_regions = num_regions * (n_samples // len(num_regions))
    tiles = []
    if split_sizes is None:
        for i, n in enumerate(num_regions):
            tiles += [i] * n
        return tiles
    split_sizes = [int(x) for x in split_sizes]
    if any(x <= 0 for x in split_sizes):
        raise RuntimeError(f"region_tile_index: every sample needs at least one tile (split_sizes = {split_sizes})")
    off = 0
    for i, (n_tiles, n) in enumerate(zip(split_sizes, num_regions)):
        if num_splits is not None:
            per_image = [int(x) for x in num_splits[i]]
            if sum(per_image) != n_tiles or any(x <= 0 for x in per_image):
                raise RuntimeError(f"region_tile_index: num_splits[{i}] = {per_image} does not add up to {n_tiles} tiles")
            if n > len(per_image):
                raise RuntimeError(f"region_tile_index: sample {i} has {n} regions but {len(per_image)} images")
            last, acc = [], 0
            for x in per_image:
                acc += x
                last.append(off + acc - 1)
            tiles += last[:n]
        else:
            tiles += [off + n_tiles - 1] * n
        off += n_tiles
    return tiles


def gather_region_images(images, num_regions, num_splits=None):
    """all_images of the region branch (modeling_visionllmv2.py:626-643): [n_all_regions, 3, h, w] -- one image per region.
    ``images``: list of per-sample tile stacks ('anyres' / mmic) or one [B, 3, h, w] tensor ('pad').  Pixels feed the region
    encoder's conv stem (torch), so this is a plain index_select."""
    if isinstance(images, (list, tuple)):
        tiles = region_tile_index([len(x) for x in images], num_regions, num_splits)
        cat = torch.cat(list(images), dim=0)
    else:
        tiles = region_tile_index(None, num_regions, n_images=images.shape[0])
        cat = images
    return cat.index_select(0, torch.as_tensor(tiles, dtype=torch.long, device=cat.device))


def gather_region_image_features(hidden_states, split_sizes, num_regions, levels=(-3, -2, -1), num_splits=None):
    """Region branch, feature selection (modeling_visionllmv2.py:644-676): for every region the features of ITS image's
    global tile without CLS, at the last three encoder levels (see region_tile_index for the three input conventions;
    ``split_sizes`` None = 'pad').  hidden_states: indexable of [n_tiles, 1 + T, C] bf16; -> list of [n_all_regions, T, C]
    (one native row gather per level)."""
    outs = []
    dev = hidden_states[levels[0]].device
    n_tiles = hidden_states[levels[0]].shape[0]
    tiles = region_tile_index(split_sizes, num_regions, num_splits, n_images=n_tiles)
    if split_sizes is not None and sum(int(x) for x in split_sizes) != n_tiles:
        raise RuntimeError(f"gather_region_image_features: split_sizes add up to {sum(split_sizes)}, the encoder saw {n_tiles} tiles")
    tile_of_region = torch.tensor(tiles, device=dev, dtype=torch.int64)
    for lv in levels:
        hs = hidden_states[lv]
        _no_grad_path("gather_region_image_features", hs)
        hs = hs.contiguous()
        n, S1, C = hs.shape
        T = S1 - 1
        src = (tile_of_region[:, None] * S1 + 1 + torch.arange(T, device=dev)[None, :]).reshape(-1)
        out = torch.empty((tile_of_region.numel(), T, C), dtype=hs.dtype, device=dev)   # every row is written: src covers all of it
        _copy_rows(hs.view(n * S1, C), src, out.view(-1, C), None, src.numel())
        outs.append(out)
    return outs


def splice_region_tokens(inputs_embeds, input_ids, reg_token_id, region_features):
    """<region> slots (modeling_visionllmv2.py:688-695): inputs_embeds[input_ids == reg_token_id] = region_features, in place."""
    B, L, C = inputs_embeds.shape
    _no_grad_path("splice_region_tokens", inputs_embeds, region_features)
    dst = torch.nonzero((input_ids == reg_token_id).reshape(-1), as_tuple=False).reshape(-1)
    feats = region_features.to(inputs_embeds.dtype).reshape(-1, C).contiguous()
    if dst.numel() != feats.shape[0]:
        raise RuntimeError(f"splice_region_tokens: {dst.numel()} <region> slots cannot take {feats.shape[0]} region features")
    _copy_rows(feats, None, inputs_embeds.view(B * L, C), dst, dst.numel())
    return inputs_embeds
