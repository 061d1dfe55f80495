"""InternViT vision encoder -- drop-in for the reference's ``InternVisionModel`` in the ``vis_encoder`` slot.

Boundary B1.  Same constructor / forward signature, config fields and PARAMETER NAMES as
VisionLLMv2/visionllmv2/model/internvit/modeling_intern_vit.py (:61-90 embeddings, :93-164 attention, :167-179 MLP,
:182-210 layer, :213-276 encoder, :279-343 model) so HF state-dicts load unchanged
(``embeddings.{class_embedding,patch_embedding.weight/bias,position_embedding}``,
``encoder.layers.{i}.{attn.qkv.weight, attn.q_norm.weight, attn.k_norm.weight, attn.proj.*, mlp.fc1/fc2.*,
norm1/norm2.weight, ls1, ls2}``).  The modules below only HOLD the parameters; ``forward`` hands their device
pointers to ``vllm_vit_forward`` (one C call for the whole encoder; hand-written HIP kernels underneath).
Inference only (the reference runs the tower under ``torch.no_grad()``, modeling_visionllmv2.py:560).
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .vit_common import (EncoderPlan, _require_bf16_cuda, fold_norm_into_linear, kpad_for, model_output, norm_folding_applies,
                         padded_patch_weight, run_encoder)

try:
    from transformers.configuration_utils import PretrainedConfig as _ConfigBase
except Exception:  # pragma: no cover
    class _ConfigBase:  # minimal stand-in
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)
            self.output_hidden_states = kw.get("output_hidden_states", False)
            self.use_return_dict = True


class InternVisionConfig(_ConfigBase):
    """Mirror of configuration_intern_vit.py:22-100 (defaults = InternViT-6B)."""
    model_type = "intern_vit_6b"

    def __init__(self, num_channels=3, patch_size=14, image_size=224, qkv_bias=False, hidden_size=3200,
                 num_attention_heads=25, intermediate_size=12800, qk_normalization=True, num_hidden_layers=48,
                 use_flash_attn=True, hidden_act="gelu", layer_norm_eps=1e-6, dropout=0.0, drop_path_rate=0.0,
                 attention_dropout=0.0, initializer_range=0.02, initializer_factor=0.1, **kwargs):
        super().__init__(**kwargs)
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.dropout = dropout
        self.drop_path_rate = drop_path_rate
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_channels = num_channels
        self.patch_size = patch_size
        self.image_size = image_size
        self.initializer_range = initializer_range
        self.initializer_factor = initializer_factor
        self.attention_dropout = attention_dropout
        self.layer_norm_eps = layer_norm_eps
        self.hidden_act = hidden_act
        self.qkv_bias = qkv_bias
        self.qk_normalization = qk_normalization
        self.use_flash_attn = use_flash_attn


class InternRMSNorm(nn.Module):
    """Parameter holder with the reference's name (modeling_intern_vit.py:33-44); also usable stand-alone."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        x = hidden_states
        _require_bf16_cuda("hidden_states", x)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty_like(x2)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().vllm_rmsnorm_bf16(_lib.ptr(x2), x2.shape[1], _lib.ptr(self.weight), _lib.ptr(y),
                                                    x2.shape[1], x2.shape[0], x2.shape[1], self.variance_epsilon,
                                                    _lib.current_stream(x.device)), "vllm_rmsnorm_bf16")
        return y.view_as(x)


class InternVisionEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_dim = config.hidden_size
        self.image_size = config.image_size
        self.patch_size = config.patch_size
        self.class_embedding = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.patch_embedding = nn.Conv2d(3, self.embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.num_patches = (self.image_size // self.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = nn.Parameter(torch.randn(1, self.num_positions, self.embed_dim))


class InternAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        if self.head_dim * self.num_heads != self.embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim} and "
                             f"`num_heads`: {self.num_heads}).")
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(self.embed_dim, 3 * self.embed_dim, bias=config.qkv_bias)
        self.qk_normalization = config.qk_normalization
        if self.qk_normalization:
            self.q_norm = InternRMSNorm(self.embed_dim, eps=config.layer_norm_eps)
            self.k_norm = InternRMSNorm(self.embed_dim, eps=config.layer_norm_eps)
        self.proj = nn.Linear(self.embed_dim, self.embed_dim)


class InternMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)


class InternVisionEncoderLayer(nn.Module):
    def __init__(self, config, drop_path_rate=0.0):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.attn = InternAttention(config)
        self.mlp = InternMLP(config)
        self.norm1 = InternRMSNorm(self.embed_dim, eps=config.layer_norm_eps)
        self.norm2 = InternRMSNorm(self.embed_dim, eps=config.layer_norm_eps)
        self.ls1 = nn.Parameter(config.initializer_factor * torch.ones(self.embed_dim))
        self.ls2 = nn.Parameter(config.initializer_factor * torch.ones(self.embed_dim))


class InternVisionEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layers = nn.ModuleList([InternVisionEncoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.gradient_checkpointing = True


class InternVisionModel(nn.Module):
    main_input_name = "pixel_values"
    config_class = InternVisionConfig

    def __init__(self, config: InternVisionConfig):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError(f"hidden_act {config.hidden_act!r}: the fused epilogue implements 'gelu' (erf)")
        self.config = config
        self.embeddings = InternVisionEmbeddings(config)
        self.encoder = InternVisionEncoder(config)
        self._plan = EncoderPlan()
        #: None = materialise every hidden state (reference behaviour); or indices to keep, e.g. (-1, -2, -3)
        self.keep_hidden_states = None

    # -- reference API ------------------------------------------------------------------------------------
    def resize_pos_embeddings(self, old_size, new_size, patch_size):
        """Same contract as modeling_intern_vit.py:291-300: the patch part of the position table is resampled bicubically
        (align_corners=False, in fp32) from the (old_size / patch)^2 grid to the (new_size / patch)^2 grid; the CLS entry is
        kept.  Host-side one-off, plain torch."""
        import torch.nn.functional as F
        table = self.embeddings.position_embedding                      # [1, 1 + g*g, C]
        g_old, g_new, C = old_size // patch_size, new_size // patch_size, table.shape[-1]
        grid = table[0, 1:].reshape(g_old, g_old, C).permute(2, 0, 1)[None].float()
        grid = F.interpolate(grid, size=(g_new, g_new), mode="bicubic", align_corners=False)
        patches = grid[0].permute(1, 2, 0).reshape(1, g_new * g_new, C).to(table.dtype)
        self.embeddings.position_embedding = nn.Parameter(torch.cat([table[:, :1], patches], dim=1))
        self.embeddings.image_size = new_size
        self.config.image_size = new_size

    def get_input_embeddings(self):
        return self.embeddings

    # -- native path ---------------------------------------------------------------------------------------
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
        emb = self.embeddings
        kpad = kpad_for(cfg.patch_size)
        pw = padded_patch_weight(emb.patch_embedding.weight, kpad)
        plan.keep = [pw]
        L = len(self.encoder.layers)
        layers = (_lib.VllmVitLayer * L)()
        P = _lib.ptr
        for i, lyr in enumerate(self.encoder.layers):
            a = lyr.attn
            fold = {}
            if norm_folding_applies(cfg.hidden_size, cfg.intermediate_size, rms=True):
                q_ln, _, q_b = fold_norm_into_linear(a.qkv.weight, a.qkv.bias, lyr.norm1.weight, None, False)
                f_ln, _, f_b = fold_norm_into_linear(lyr.mlp.fc1.weight, lyr.mlp.fc1.bias, lyr.norm2.weight, None, False)
                plan.keep += [q_ln, q_b, f_ln, f_b]
                fold = dict(qkv_w_ln=P(q_ln), qkv_colsum=None, qkv_bias_ln=P(q_b) if q_b is not None else None, fc1_w_ln=P(f_ln),
                            fc1_colsum=None, fc1_bias_ln=P(f_b) if f_b is not None else None)
            layers[i] = _lib.VllmVitLayer(
                norm1_w=P(lyr.norm1.weight), norm1_b=None, qkv_w=P(a.qkv.weight), qkv_b=P(a.qkv.bias),
                q_norm_w=P(a.q_norm.weight) if a.qk_normalization else None,
                k_norm_w=P(a.k_norm.weight) if a.qk_normalization else None,
                proj_w=P(a.proj.weight), proj_b=P(a.proj.bias), ls1=P(lyr.ls1), norm2_w=P(lyr.norm2.weight),
                norm2_b=None, fc1_w=P(lyr.mlp.fc1.weight), fc1_b=P(lyr.mlp.fc1.bias), fc2_w=P(lyr.mlp.fc2.weight),
                fc2_b=P(lyr.mlp.fc2.bias), ls2=P(lyr.ls2), **fold)
        desc = _lib.VllmVitDesc(
            arch=_lib.ARCH_INTERNVIT, num_layers=L, hidden=cfg.hidden_size, heads=cfg.num_attention_heads,
            inter=cfg.intermediate_size, patch=cfg.patch_size,
            # the position table decides the tile size (resize_pos_embeddings may have changed it)
            image=int(round((emb.position_embedding.shape[1] - 1) ** 0.5)) * cfg.patch_size, kpad=kpad,
            act=_lib.EPI_GELU,
            pixel_is_f32=0, eps=cfg.layer_norm_eps, patch_w=P(pw), patch_b=P(emb.patch_embedding.bias),
            cls=P(emb.class_embedding), pos=P(emb.position_embedding), pre_ln_w=None, pre_ln_b=None,
            layers=ctypes.cast(layers, ctypes.POINTER(_lib.VllmVitLayer)))
        plan.key, plan.desc, plan.layers = key, desc, layers
        return desc

    @torch.no_grad()
    def forward(self, pixel_values=None, output_hidden_states=None, return_dict=None, pixel_embeds=None):
        cfg = self.config
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else getattr(cfg, "output_hidden_states", False))
        return_dict = return_dict if return_dict is not None else getattr(cfg, "use_return_dict", True)
        if pixel_values is None and pixel_embeds is None:
            raise ValueError("You have to specify pixel_values or pixel_embeds")
        if pixel_embeds is not None:
            raise NotImplementedError("pixel_embeds input is not wired to the native encoder (unused by VisionLLMv2)")
        if len(pixel_values.shape) != 4:
            raise ValueError(f"wrong pixel_values size: {pixel_values.shape}")
        desc = self._build_plan()
        keep = self.keep_hidden_states if output_hidden_states else (-1,)
        states = run_encoder(desc, pixel_values, desc.num_layers, cfg.hidden_size, keep)
        last = states[-1]
        hs = tuple(states) if output_hidden_states else None
        return model_output(last, last[:, 0, :], hs, return_dict)
