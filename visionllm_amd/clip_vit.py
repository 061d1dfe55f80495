"""CLIP ViT (ViT-L/14-336 in the released 7B model) -- drop-in for ``transformers.CLIPVisionModel`` in the
``vis_encoder`` slot (VisionLLMv2/visionllmv2/model/modeling_visionllmv2.py:132-135; train/train.py:382-384).

The arithmetic is third-party (transformers, pinned ==4.34.0 in requirements.txt:22).  Parameter names follow the
HF 4.34 layout the reference's checkpoints use (``vision_model.embeddings.{class_embedding, patch_embedding.weight,
position_embedding.weight}``, ``vision_model.pre_layrnorm.*``, ``vision_model.encoder.layers.{i}.{self_attn.
{q,k,v,out}_proj.*, layer_norm1/2.*, mlp.fc1/fc2.*}``, ``vision_model.post_layernorm.*``); ``load_state_dict`` also
accepts the prefix-less layout newer transformers versions emit.  The modules only hold parameters; ``forward``
calls ``vllm_vit_forward`` (q/k/v projections run as ONE fused [3C,C] GEMM).
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .vit_common import (EncoderPlan, _require_bf16_cuda, fold_norm_into_linear, kpad_for, model_output, norm_folding_applies,
                         padded_patch_weight, run_encoder)

try:
    from transformers import CLIPVisionConfig  # noqa: F401  (the reference passes this very config class)
except Exception:  # pragma: no cover
    CLIPVisionConfig = None


class _CLIPVisionEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.class_embedding = nn.Parameter(torch.randn(self.embed_dim))
        self.patch_embedding = nn.Conv2d(3, self.embed_dim, kernel_size=self.patch_size, stride=self.patch_size,
                                         bias=False)
        self.num_patches = (self.image_size // self.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = nn.Embedding(self.num_positions, self.embed_dim)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)), persistent=False)


class _CLIPAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        C = config.hidden_size
        self.k_proj = nn.Linear(C, C)
        self.v_proj = nn.Linear(C, C)
        self.q_proj = nn.Linear(C, C)
        self.out_proj = nn.Linear(C, C)


class _CLIPMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)


class _CLIPEncoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = _CLIPAttention(config)
        self.layer_norm1 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.mlp = _CLIPMLP(config)
        self.layer_norm2 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _CLIPEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([_CLIPEncoderLayer(config) for _ in range(config.num_hidden_layers)])


class _CLIPVisionTransformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embeddings = _CLIPVisionEmbeddings(config)
        self.pre_layrnorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.encoder = _CLIPEncoder(config)
        self.post_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class CLIPVisionModel(nn.Module):
    main_input_name = "pixel_values"

    def __init__(self, config):
        super().__init__()
        act = getattr(config, "hidden_act", "quick_gelu")
        if act not in ("quick_gelu", "gelu"):
            raise NotImplementedError(f"hidden_act {act!r}")
        self.config = config
        self.vision_model = _CLIPVisionTransformer(config)
        self._plan = EncoderPlan()
        self.keep_hidden_states = None

    def get_input_embeddings(self):
        return self.vision_model.embeddings.patch_embedding

    def load_state_dict(self, state_dict, strict=True, **kw):
        if not any(k.startswith("vision_model.") for k in state_dict):
            state_dict = {"vision_model." + k: v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _build_plan(self):
        cfg = self.config
        params = list(self.parameters())
        key = EncoderPlan.signature(params)
        plan = self._plan
        if plan.key == key:
            return plan.desc
        for n_, p in self.named_parameters():
            _require_bf16_cuda(n_, p)
        _lib.check_struct_layouts()
        vm = self.vision_model
        emb = vm.embeddings
        kpad = kpad_for(cfg.patch_size)
        pw = padded_patch_weight(emb.patch_embedding.weight, kpad)
        plan.keep = [pw]
        L = len(vm.encoder.layers)
        layers = (_lib.VllmVitLayer * L)()
        P = _lib.ptr
        for i, lyr in enumerate(vm.encoder.layers):
            a = lyr.self_attn
            qkv_w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach().contiguous()
            qkv_b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).detach().contiguous()
            plan.keep += [qkv_w, qkv_b]
            fold = {}
            if norm_folding_applies(cfg.hidden_size, cfg.intermediate_size):
                q_ln, q_cs, q_b = fold_norm_into_linear(qkv_w, qkv_b, lyr.layer_norm1.weight, lyr.layer_norm1.bias, True)
                f_ln, f_cs, f_b = fold_norm_into_linear(lyr.mlp.fc1.weight, lyr.mlp.fc1.bias, lyr.layer_norm2.weight,
                                                        lyr.layer_norm2.bias, True)
                plan.keep += [q_ln, q_cs, q_b, f_ln, f_cs, f_b]
                fold = dict(qkv_w_ln=P(q_ln), qkv_colsum=P(q_cs), qkv_bias_ln=P(q_b), fc1_w_ln=P(f_ln), fc1_colsum=P(f_cs),
                            fc1_bias_ln=P(f_b))
            layers[i] = _lib.VllmVitLayer(
                norm1_w=P(lyr.layer_norm1.weight), norm1_b=P(lyr.layer_norm1.bias), qkv_w=P(qkv_w), qkv_b=P(qkv_b),
                q_norm_w=None, k_norm_w=None, proj_w=P(a.out_proj.weight), proj_b=P(a.out_proj.bias), ls1=None,
                norm2_w=P(lyr.layer_norm2.weight), norm2_b=P(lyr.layer_norm2.bias), fc1_w=P(lyr.mlp.fc1.weight),
                fc1_b=P(lyr.mlp.fc1.bias), fc2_w=P(lyr.mlp.fc2.weight), fc2_b=P(lyr.mlp.fc2.bias), ls2=None, **fold)
        act = _lib.EPI_QUICK_GELU if getattr(cfg, "hidden_act", "quick_gelu") == "quick_gelu" else _lib.EPI_GELU
        desc = _lib.VllmVitDesc(
            arch=_lib.ARCH_CLIP, num_layers=L, hidden=cfg.hidden_size, heads=cfg.num_attention_heads,
            inter=cfg.intermediate_size, patch=cfg.patch_size, image=cfg.image_size, kpad=kpad, act=act, pixel_is_f32=0,
            eps=cfg.layer_norm_eps, patch_w=P(pw), patch_b=None, cls=P(emb.class_embedding),
            pos=P(emb.position_embedding.weight), pre_ln_w=P(vm.pre_layrnorm.weight), pre_ln_b=P(vm.pre_layrnorm.bias),
            layers=ctypes.cast(layers, ctypes.POINTER(_lib.VllmVitLayer)))
        plan.key, plan.desc, plan.layers = key, desc, layers
        return desc

    @torch.no_grad()
    def forward(self, pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None, **kw):
        cfg = self.config
        if output_attentions:
            raise NotImplementedError("attention maps are never materialised by the fused attention kernel")
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(cfg, "output_hidden_states", False))
        return_dict = return_dict if return_dict is not None else getattr(cfg, "use_return_dict", True)
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        desc = self._build_plan()
        keep = self.keep_hidden_states if output_hidden_states else (-1,)
        states = run_encoder(desc, pixel_values, desc.num_layers, cfg.hidden_size, keep)
        last = states[-1]
        # pooler_output = post_layernorm(last[:, 0])  (HF CLIPVisionTransformer); unused by VisionLLMv2 but kept
        n, S, C = last.shape
        pooled = torch.empty((n, C), dtype=last.dtype, device=last.device)
        pl = self.vision_model.post_layernorm
        with torch.cuda.device(last.device):
            _lib.check(_lib.lib().vllm_layernorm_bf16(_lib.ptr(last), S * C, _lib.ptr(pl.weight), _lib.ptr(pl.bias),
                                                      _lib.ptr(pooled), C, n, C, pl.eps,
                                                      _lib.current_stream(last.device)), "vllm_layernorm_bf16")
        hs = tuple(states) if output_hidden_states else None
        return model_output(last, pooled, hs, return_dict)
