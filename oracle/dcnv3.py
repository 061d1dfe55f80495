"""ORACLE -- test infrastructure only.  ctypes front-end of oracle/dcnv3_oracle.c (DCNv3 forward + backward) plus a torch
restatement of the reference's pure-PyTorch twin ``dcnv3_core_pytorch``
(VisionLLMv2/visionllmv2/model/ops_dcnv3/functions/dcnv3_func.py:61-161), which is the path the reference runs on CPU."""
import ctypes
import os

import numpy as np

from . import build

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libdcnv3_oracle.so"))
    return _lib


def out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    """dcnv3_cuda.cu:40-45"""
    return (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1


def forward(inp, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale):
    dt = np.float64 if inp.dtype == np.float64 else np.float32
    inp, offset, mask = (np.ascontiguousarray(a, dt) for a in (inp, offset, mask))
    N, H, W, C = inp.shape
    assert C == group * group_channels
    Ho, Wo = out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert offset.shape == (N, Ho, Wo, group * kh * kw * 2) and mask.shape == (N, Ho, Wo, group * kh * kw)
    out = np.empty((N, Ho, Wo, C), dt)
    fn = lib().dcnv3_forward_f64 if dt == np.float64 else lib().dcnv3_forward_f32
    sc = ctypes.c_double(offset_scale) if dt == np.float64 else ctypes.c_float(offset_scale)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn(p(inp), p(offset), p(mask), N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw, sc, Ho, Wo, p(out))
    return out


def backward(inp, offset, mask, grad_out, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale):
    """-> (grad_input, grad_offset, grad_mask): the reference's dcnv3_col2im (dcnv3_im2col_cuda.cuh:86-146, 279-857) restated."""
    dt = np.float64 if inp.dtype == np.float64 else np.float32
    inp, offset, mask, grad_out = (np.ascontiguousarray(a, dt) for a in (inp, offset, mask, grad_out))
    N, H, W, C = inp.shape
    assert C == group * group_channels
    Ho, Wo = out_size(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert grad_out.shape == (N, Ho, Wo, C)
    gi, go, gm = np.zeros_like(inp), np.empty_like(offset), np.empty_like(mask)
    fn = lib().dcnv3_backward_f64 if dt == np.float64 else lib().dcnv3_backward_f32
    sc = ctypes.c_double(offset_scale) if dt == np.float64 else ctypes.c_float(offset_scale)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn(p(inp), p(offset), p(mask), p(grad_out), N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw, sc, Ho, Wo,
       p(gi), p(go), p(gm))
    return gi, go, gm


def core_pytorch_twin(inp, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale):
    """torch restatement of dcnv3_core_pytorch (pad, reference points :61-91, dilation grid :94-118, grid_sample :121-161)."""
    import torch
    import torch.nn.functional as F
    inp = F.pad(inp, [0, 0, ph, ph, pw, pw])
    N, H, W, _ = inp.shape
    _, Ho, Wo, _ = offset.shape
    dev = inp.device
    ys = torch.linspace((dh * (kh - 1)) // 2 + 0.5, (dh * (kh - 1)) // 2 + 0.5 + (Ho - 1) * sh, Ho, dtype=torch.float32,
                        device=dev)
    xs = torch.linspace((dw * (kw - 1)) // 2 + 0.5, (dw * (kw - 1)) // 2 + 0.5 + (Wo - 1) * sw, Wo, dtype=torch.float32,
                        device=dev)
    ry, rx = torch.meshgrid(ys, xs, indexing="ij")
    ref = torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1).reshape(1, Ho, Wo, 1, 2)
    gx, gy = torch.meshgrid(
        torch.linspace(-((dw * (kw - 1)) // 2), -((dw * (kw - 1)) // 2) + (kw - 1) * dw, kw, dtype=torch.float32, device=dev),
        torch.linspace(-((dh * (kh - 1)) // 2), -((dh * (kh - 1)) // 2) + (kh - 1) * dh, kh, dtype=torch.float32, device=dev),
        indexing="ij")
    grid = torch.stack([gx / W, gy / H], -1).reshape(-1, 1, 2).repeat(1, group, 1).permute(1, 0, 2)
    grid = grid.reshape(1, 1, 1, group * kh * kw, 2)
    norm = torch.tensor([W, H], device=dev).reshape(1, 1, 1, 2).repeat(1, 1, 1, group * kh * kw)
    loc = (ref + grid * offset_scale).repeat(N, 1, 1, 1, 1).flatten(3, 4) + offset * offset_scale / norm
    P = kh * kw
    grids = 2 * loc - 1
    x = inp.view(N, H * W, group * group_channels).transpose(1, 2).reshape(N * group, group_channels, H, W)
    g = grids.view(N, Ho * Wo, group, P, 2).transpose(1, 2).flatten(0, 1)
    s = F.grid_sample(x, g, mode="bilinear", padding_mode="zeros", align_corners=False)
    m = mask.view(N, Ho * Wo, group, P).transpose(1, 2).reshape(N * group, 1, Ho * Wo, P)
    out = (s * m).sum(-1).view(N, group * group_channels, Ho * Wo)
    return out.transpose(1, 2).reshape(N, Ho, Wo, -1).contiguous()
