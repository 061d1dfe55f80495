"""Region encoder (visual-prompt masks -> one embedding per region): host-side mirror, SURVEY.md section 8 row f4.

Same names and parameter layout as visionllmv2/model/region_encoder.py: ``LayerNorm2d`` (:9-21), ``point_sample``
(:24-47), ``rand_sample`` (:50-65), ``RegionEncoder`` (:68-147).  The convolutional mask embedding stays torch modules
(MIOpen); the per-region pooling of the 'grid_sample' type -- bilinear sampling of up to 2304 points per region and their
masked mean -- is ONE native call (libvllm_hip.so: vllm_point_sample_mean_f32) instead of grid_sample + mask + sum + div.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


def _check_ps(input, point_coords, kwargs):
    if kwargs.get("mode", "bilinear") != "bilinear" or kwargs.get("padding_mode", "zeros") != "zeros" or \
            kwargs.get("align_corners", False):
        raise NotImplementedError("point_sample: the native kernel implements bilinear / zeros / align_corners=False "
                                  "(the only combination the reference uses, region_encoder.py:135)")
    if not input.is_cuda:
        raise RuntimeError("point_sample: input must be a CUDA tensor (no CPU fallback)")
    if input.dim() != 4 or point_coords.shape[-1] != 2 or point_coords.shape[0] != input.shape[0]:
        raise ValueError("point_sample: input must be (N, C, H, W) and point_coords (N, P, 2) or (N, Hg, Wg, 2)")


def point_sample(input, point_coords, **kwargs):
    """(N, C, H, W), (N, P, 2) or (N, Hg, Wg, 2) in [0, 1]^2 -> (N, C, P) or (N, C, Hg, Wg)."""
    _check_ps(input, point_coords, kwargs)
    N, C, H, W = input.shape
    grid = point_coords.dim() == 4
    pts = point_coords.reshape(N, -1, 2).to(torch.float32).contiguous()
    x = input.to(torch.float32).contiguous()
    P = pts.shape[1]
    out = torch.empty((N, C, P), dtype=torch.float32, device=x.device)
    if torch.is_grad_enabled() and (input.requires_grad or point_coords.requires_grad):
        raise RuntimeError("point_sample (native): forward-only kernel -- call under torch.no_grad(); the reference "
                           "differentiates through F.grid_sample here")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vllm_point_sample_f32(_lib.ptr(x), _lib.ptr(pts), N, C, H, W, P, _lib.ptr(out),
                                                    _lib.current_stream(x.device)), "vllm_point_sample_f32")
    out = out.to(input.dtype)
    return out.reshape(N, C, point_coords.shape[1], point_coords.shape[2]) if grid else out


def point_sample_masked_mean(input, point_coords, valid):
    """Fused form of region_encoder.py:135-140: mean over the valid points of every region, (N, C)."""
    _check_ps(input, point_coords, {})
    N, C, H, W = input.shape
    pts = point_coords.to(torch.float32).contiguous()
    P = pts.shape[1]
    x = input.to(torch.float32).contiguous()
    v = valid.to(torch.uint8).contiguous()
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    if torch.is_grad_enabled() and (input.requires_grad or point_coords.requires_grad):
        # region_encoder.py:135-140 trains mask_embedding through grid_sample; this kernel has no backward
        raise RuntimeError("point_sample_masked_mean (native): forward-only kernel -- call under torch.no_grad()")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vllm_point_sample_mean_f32(_lib.ptr(x), _lib.ptr(pts), _lib.ptr(v), N, C, H, W, P, _lib.ptr(out),
                                                         _lib.current_stream(x.device)), "vllm_point_sample_mean_f32")
    return out.to(input.dtype)


def rand_sample(x, divisor, max_len):
    """region_encoder.py:50-65: up to ``max_len`` random non-zero positions of a region mask ``x`` [n_masks, H, W], each
    mask channel carrying the same probability mass; rows of the result are (mask id, y / H, x / W), sorted by position."""
    idx = x.nonzero()
    if idx.shape[0] == 0:
        return idx
    pts = idx / divisor                                   # [n, 3]
    ids = idx[:, 0]
    counts = torch.bincount(ids)
    mass = 1.0 / ((counts > 0).sum() * counts[ids])       # 1 / (masks present * points of this mask)
    keep = torch.multinomial(mass.to(pts.dtype), num_samples=min(max_len, mass.numel()), replacement=False).sort()[0]
    return pts[keep]


def _mask_stem(hidden_dim, embed_dim, kernel_size):
    """4-channel (RGB + mask) stem down to the ViT patch grid: stride patch/2, then stride 2 (region_encoder.py:76-84)."""
    return nn.Sequential(
        nn.Conv2d(4, hidden_dim // 4, kernel_size=kernel_size, stride=kernel_size),
        LayerNorm2d(hidden_dim // 4),
        nn.GELU(),
        nn.Conv2d(hidden_dim // 4, hidden_dim, kernel_size=2, stride=2),
        LayerNorm2d(hidden_dim),
        nn.GELU(),
        nn.Conv2d(hidden_dim, embed_dim, kernel_size=1),
    )


class RegionEncoder(nn.Module):
    """region_encoder.py:68-147 with the reference's parameter names (mask_embedding.*, region_query, region_attn.*,
    up_dim.*)."""

    POOL_TYPES = ("mean", "cross_attn", "grid_sample")

    def __init__(self, hidden_dim, embed_dim, out_dim, patch_size=14, mask_pool_type="mean"):
        super().__init__()
        if patch_size % 2 != 0:
            raise AssertionError("patch_size must be even (the stem strides by patch_size / 2, then by 2)")
        if mask_pool_type not in self.POOL_TYPES:
            raise AssertionError(f"mask_pool_type must be one of {self.POOL_TYPES}")
        self.patch_size = patch_size
        self.mask_pool_type = mask_pool_type
        self.mask_embedding = _mask_stem(hidden_dim, embed_dim, patch_size // 2)
        if mask_pool_type == "cross_attn":
            self.region_query = nn.Embedding(1, embed_dim)
            self.region_attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=8, dropout=0.0, batch_first=True)
        elif mask_pool_type == "grid_sample":
            self.num_points = 2304   # 48 x 48
        self.up_dim = nn.Linear(embed_dim, out_dim)

    # -- the three pooling flavours (:117-141) --
    def _pool_mean(self, feat, masks):
        h, w = feat.shape[-2:]
        inside = F.interpolate(masks.float(), size=(h, w), mode="bilinear", align_corners=False) > 0.5
        feat = feat * inside
        return feat, feat.mean(-1).mean(-1)

    def _pool_attn(self, feat):
        kv = feat.flatten(-2).transpose(1, 2)
        q = self.region_query.weight.unsqueeze(0).repeat(feat.shape[0], 1, 1)
        return kv, self.region_attn(q, kv, kv)[0].squeeze(1)

    def _pool_points(self, feat, masks):
        H, W = masks.shape[-2:]
        divisor = torch.tensor([1, H, W], device=masks.device)[None,]
        pts = nn.utils.rnn.pad_sequence([rand_sample(m, divisor, self.num_points) for m in masks], padding_value=-1)
        pts = pts.permute(1, 0, 2)                        # [regions, points, (id, y, x)], -1 padded
        valid = pts.sum(dim=-1) >= 0
        return point_sample_masked_mean(feat, pts[:, :, -2:].flip(dims=[-1]), valid)   # native: sample + masked mean

    def forward(self, images, masks, image_features):
        assert images.shape[-2:] == masks.shape[-2:]
        masks = masks.to(images.dtype)
        feat = self.mask_embedding(torch.cat([images, masks], dim=1))
        n, (h, w) = len(images), feat.shape[-2:]
        pooled = []
        for level in image_features:
            if level.dim() == 3:
                level = level.reshape(n, h, w, -1).permute(0, 3, 1, 2)
            assert feat.shape[-2:] == level.shape[-2:]
            feat = feat + level
            if self.mask_pool_type == "mean":
                feat, region = self._pool_mean(feat, masks)      # (the reference carries the masked map to the next level)
            elif self.mask_pool_type == "cross_attn":
                feat, region = self._pool_attn(feat)             # (and the flattened one here)
            else:
                region = self._pool_points(feat, masks)
            pooled.append(self.up_dim(region))
        return torch.stack(pooled).mean(dim=0)
