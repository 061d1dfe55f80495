"""Visual-token splice: writes the projector output into the ``<im_patch>`` slots of the LLM input embeddings
(VisionLLMv2/visionllmv2/model/modeling_visionllmv2.py:582-605).  The reference does it with a boolean-mask assignment
``inputs_embeds[selected] = inputs_embeds[selected] * 0.0 + vit_embeds``; here the slot positions come from
``torch.nonzero`` (index bookkeeping) and the row movement is one HIP scatter kernel, in place."""
import torch

from . import _lib


def splice_visual_tokens(inputs_embeds, input_ids, imp_token_id, image_features, split_sizes=None):
    """inputs_embeds [B, L, C] (bf16, CUDA, modified in place and returned), input_ids [B, L],
    image_features [n_tiles, T, C] in tile order, split_sizes: tiles per sample ('anyres' list input) or None.

    Mirrors the reference's handling of samples without an image (their tiles are dropped, :585-592) and of a
    token-count mismatch (:597-603): features are repeated when the slots are a whole multiple of them; any other mismatch
    raises, as the reference's second assignment does."""
    B, L, C = inputs_embeds.shape
    if not inputs_embeds.is_cuda or inputs_embeds.dtype != torch.bfloat16 or not inputs_embeds.is_contiguous():
        raise RuntimeError("splice_visual_tokens: inputs_embeds must be a contiguous bf16 CUDA tensor")
    selected = input_ids == imp_token_id
    has_image = selected.sum(-1) != 0
    if split_sizes is not None:
        has_image = torch.cat([has_image[i][None].repeat(int(split_sizes[i])) for i in range(B)], dim=0)
    vit = image_features[has_image].reshape(-1, C).to(inputs_embeds.dtype).contiguous()
    idx = torch.nonzero(selected.reshape(-1), as_tuple=False).reshape(-1)
    n_sel, n_vit = idx.numel(), vit.shape[0]
    if n_sel != n_vit:
        if n_vit > 0 and n_sel > n_vit and n_sel % n_vit == 0:
            vit = vit.repeat(n_sel // n_vit, 1)
        else:
            raise RuntimeError(f"splice_visual_tokens: shape mismatch: {n_sel} <im_patch> slots cannot take {n_vit} visual tokens")
    n = idx.numel()
    if n:
        with torch.cuda.device(inputs_embeds.device):
            _lib.check(_lib.lib().vllm_scatter_rows_bf16(_lib.ptr(vit), _lib.ptr(idx.contiguous()), _lib.ptr(inputs_embeds),
                                                         n, C, B * L, _lib.current_stream(inputs_embeds.device)),
                       "vllm_scatter_rows_bf16")
    return inputs_embeds
