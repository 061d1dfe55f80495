"""Bring-up hook B4: drop-in for the reference's ``FlashAttention`` module
(VisionLLMv2/visionllmv2/model/internvit/flash_attention.py:14-76), backed by ``vllm_attn_fwd_qkvpacked_bf16`` / ``_f16``.

Only the call pattern the vision tower uses is supported -- ``forward(qkv[B,S,3,H,D], key_padding_mask=None,
causal=False)`` in eval mode (``modeling_intern_vit.py:155-157``); anything else raises instead of silently computing
something different."""
import torch
from torch import nn

from . import _lib


class FlashAttention(nn.Module):
    def __init__(self, softmax_scale=None, attention_dropout=0.0, device=None, dtype=None):
        super().__init__()
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None, need_weights=False):
        assert not need_weights
        if key_padding_mask is not None or causal or cu_seqlens is not None:
            raise NotImplementedError("native FlashAttention: only dense non-causal qkv[B,S,3,H,D] (the ViT tile case)")
        if self.training and self.dropout_p > 0:
            raise NotImplementedError("native FlashAttention: attention dropout is not implemented (inference path)")
        # the reference accepts fp16 and bf16 (flash_attention.py:39-41: `assert qkv.dtype in [torch.float16, torch.bfloat16]`)
        if qkv.dtype not in (torch.bfloat16, torch.float16) or not qkv.is_cuda or qkv.dim() != 5 or qkv.shape[2] != 3:
            raise RuntimeError("native FlashAttention: qkv must be a bf16 / fp16 CUDA tensor [B, S, 3, H, D] (flash_attention.py:39-41)")
        qkv = qkv.contiguous()
        B, S, _, H, D = qkv.shape
        out = torch.empty((B, S, H, D), dtype=qkv.dtype, device=qkv.device)
        scale = self.softmax_scale if self.softmax_scale is not None else D ** -0.5
        with torch.cuda.device(qkv.device):
            fn = _lib.lib().vllm_attn_fwd_qkvpacked_f16 if qkv.dtype == torch.float16 else _lib.lib().vllm_attn_fwd_qkvpacked_bf16
            _lib.check(fn(_lib.ptr(qkv), _lib.ptr(out), B, S, H, D, float(scale), _lib.current_stream(qkv.device)),
                       "vllm_attn_fwd_qkvpacked")
        return out, None
