"""DCNv3 (deformable convolution v3 of the InternImage det backbone): host-side mirror of the reference operator.

SURVEY.md section 8 row f3.  Same names, argument order and error behaviour as

* ``DCNv3.dcnv3_forward``  -- the pybind module of visionllmv2/model/ops_dcnv3/src/vision.cpp (signature src/dcnv3.h:20-39);
* ``DCNv3Function``        -- visionllmv2/model/ops_dcnv3/functions/dcnv3_func.py:21-59 (forward and, since round 4, backward:
  ``dcnv3_backward`` -> ``vllm_dcnv3_backward_f32 / _f64``);
* ``DCNv3`` (module)       -- visionllmv2/model/ops_dcnv3/modules/dcnv3.py:222-349: the projections, depth-wise conv,
  norm / activation and softmax stay torch modules with the reference's parameter names, the sampling core is the native
  kernel (fp32, as the reference upcasts around it, :330-340).

The compute is libvllm_hip.so (visionllm_amd/csrc/dcnv3.hip); like the reference's extension ("Not implement on cpu",
src/cpu/dcnv3_cpu.cpp:25) it raises for CPU tensors.
"""
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import constant_, xavier_uniform_

from . import _lib


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                  group, group_channels, offset_scale, im2col_step=256):
    """input [N, H, W, group*group_channels], offset [N, Ho, Wo, group*kh*kw*2], mask [N, Ho, Wo, group*kh*kw] (fp16, fp32 or
    fp64, contiguous, on the GPU; fp16: half operands, fp32 arithmetic, one rounding of the result) -> [N, Ho, Wo, group*group_channels].  ``im2col_step`` is checked as the reference
    does (batch must be divisible by min(batch, im2col_step), dcnv3_cuda.cu:46-49) and otherwise unused (one launch)."""
    for name, t in (("input", input), ("offset", offset), ("mask", mask)):
        if not t.is_cuda:
            raise RuntimeError("Not implement on cpu ({} must be a CUDA tensor)".format(name))
        if not t.is_contiguous():
            raise RuntimeError("{} tensor has to be contiguous".format(name))
    if input.dtype not in (torch.float16, torch.float32, torch.float64) or offset.dtype != input.dtype or mask.dtype != input.dtype:
        raise RuntimeError("dcnv3_forward: input, offset and mask must share dtype float16, float32 or float64 "
                           "(AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69)")
    N, H, W, C = input.shape
    if C != group * group_channels:
        raise RuntimeError("Input channels and group times group channels wont match: ({} vs {}).".format(
            C, group * group_channels))
    step = min(N, im2col_step)
    if N > 0 and N % step != 0:
        raise RuntimeError("batch({}) must divide im2col_step({})".format(N, step))
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    P = kernel_h * kernel_w
    if tuple(offset.shape) != (N, Ho, Wo, group * P * 2) or tuple(mask.shape) != (N, Ho, Wo, group * P):
        raise RuntimeError("dcnv3_forward: offset / mask do not match the output geometry [{}, {}, {}, .]".format(N, Ho, Wo))
    out = torch.empty((N, Ho, Wo, C), dtype=input.dtype, device=input.device)
    L = _lib.lib()
    fn = {torch.float16: L.vllm_dcnv3_forward_f16, torch.float32: L.vllm_dcnv3_forward_f32, torch.float64: L.vllm_dcnv3_forward_f64}[input.dtype]
    with torch.cuda.device(input.device):
        _lib.check(fn(_lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), N, H, W, group, group_channels, kernel_h, kernel_w,
                      stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, float(offset_scale), _lib.ptr(out),
                      _lib.current_stream(input.device)), "vllm_dcnv3_forward")
    return out


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                   group, group_channels, offset_scale, grad_output, im2col_step=256):
    """DCNv3.dcnv3_backward (ops_dcnv3/src/dcnv3.h:41-64; functions/dcnv3_func.py:51-59): -> (grad_input, grad_offset, grad_mask),
    shaped like input / offset / mask.  Same checks as the forward; ``im2col_step`` is validated and otherwise unused."""
    for name, t in (("input", input), ("offset", offset), ("mask", mask), ("grad_output", grad_output)):
        if not t.is_cuda:
            raise RuntimeError("Not implement on cpu ({} must be a CUDA tensor)".format(name))
        if not t.is_contiguous():
            raise RuntimeError("{} tensor has to be contiguous".format(name))
    if input.dtype not in (torch.float16, torch.float32, torch.float64) or any(t.dtype != input.dtype for t in (offset, mask, grad_output)):
        raise RuntimeError("dcnv3_backward: input, offset, mask and grad_output must share dtype float16, float32 or float64")
    N, H, W, C = input.shape
    if C != group * group_channels:
        raise RuntimeError("Input channels and group times group channels wont match: ({} vs {}).".format(
            C, group * group_channels))
    step = min(N, im2col_step)
    if N > 0 and N % step != 0:
        raise RuntimeError("batch({}) must divide im2col_step({})".format(N, step))
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    P = kernel_h * kernel_w
    if tuple(offset.shape) != (N, Ho, Wo, group * P * 2) or tuple(mask.shape) != (N, Ho, Wo, group * P) or \
            tuple(grad_output.shape) != (N, Ho, Wo, C):
        raise RuntimeError("dcnv3_backward: offset / mask / grad_output do not match the output geometry [{}, {}, {}, .]".format(N, Ho, Wo))
    L = _lib.lib()
    if input.dtype == torch.float16:
        # half operands, fp32 arithmetic on widened copies in a workspace this call owns, every gradient rounded once (the native
        # entry point does the widening / zero fill / narrowing; dcnv3_cuda.cu:147 dispatches AND_HALF)
        grad_input, grad_offset, grad_mask = torch.empty_like(input), torch.empty_like(offset), torch.empty_like(mask)
        nbytes = L.vllm_dcnv3_backward_f16_workspace(N, H, W, group, group_channels, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                                     dilation_h, dilation_w)
        if nbytes < 0:
            raise RuntimeError("dcnv3_backward: invalid geometry")
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=input.device)
        with torch.cuda.device(input.device):
            _lib.check(L.vllm_dcnv3_backward_f16(_lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(grad_output), N, H, W, group,
                                                 group_channels, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                                 float(offset_scale), _lib.ptr(grad_input), _lib.ptr(grad_offset), _lib.ptr(grad_mask),
                                                 _lib.ptr(ws), nbytes, _lib.current_stream(input.device)), "vllm_dcnv3_backward_f16")
        return grad_input, grad_offset, grad_mask
    grad_input = torch.zeros_like(input)      # (the sums arrive by atomics: dcnv3_cuda.cu:118 zero-fills it as well)
    grad_offset = torch.empty_like(offset)
    grad_mask = torch.empty_like(mask)
    fn = L.vllm_dcnv3_backward_f32 if input.dtype == torch.float32 else L.vllm_dcnv3_backward_f64
    with torch.cuda.device(input.device):
        _lib.check(fn(_lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(grad_output), N, H, W, group, group_channels,
                      kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, float(offset_scale),
                      _lib.ptr(grad_input), _lib.ptr(grad_offset), _lib.ptr(grad_mask), _lib.current_stream(input.device)),
                   "vllm_dcnv3_backward")
    return grad_input, grad_offset, grad_mask


class DCNv3Function(Function):
    """functions/dcnv3_func.py:21-59 (forward :23-49, backward :51-59)."""

    @staticmethod
    def forward(ctx, input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                group, group_channels, offset_scale, im2col_step):
        ctx.geo = (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                   offset_scale, im2col_step)
        input, offset, mask = input.contiguous(), offset.contiguous(), mask.contiguous()
        ctx.save_for_backward(input, offset, mask)
        return dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                             group, group_channels, offset_scale, im2col_step)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask = ctx.saved_tensors
        kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale, im2col_step = ctx.geo
        grad_input, grad_offset, grad_mask = dcnv3_backward(input, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group,
                                                            group_channels, offset_scale, grad_output.contiguous(), im2col_step)
        return (grad_input, grad_offset, grad_mask) + (None,) * 12


class to_channels_first(nn.Module):
    def forward(self, x):
        return x.permute(0, 3, 1, 2)


class to_channels_last(nn.Module):
    def forward(self, x):
        return x.permute(0, 2, 3, 1)


def build_norm_layer(dim, norm_layer, in_format="channels_last", out_format="channels_last", eps=1e-6):
    layers = []
    if norm_layer == "BN":
        if in_format == "channels_last":
            layers.append(to_channels_first())
        layers.append(nn.BatchNorm2d(dim))
        if out_format == "channels_last":
            layers.append(to_channels_last())
    elif norm_layer == "LN":
        if in_format == "channels_first":
            layers.append(to_channels_last())
        layers.append(nn.LayerNorm(dim, eps=eps))
        if out_format == "channels_first":
            layers.append(to_channels_first())
    else:
        raise NotImplementedError(f"build_norm_layer does not support {norm_layer}")
    return nn.Sequential(*layers)


def build_act_layer(act_layer):
    if act_layer == "ReLU":
        return nn.ReLU(inplace=True)
    if act_layer == "SiLU":
        return nn.SiLU(inplace=True)
    if act_layer == "GELU":
        return nn.GELU()
    raise NotImplementedError(f"build_act_layer does not support {act_layer}")


class DCNv3(nn.Module):
    """modules/dcnv3.py:222-349 with the reference's parameter names (dw_conv.*, offset.*, mask.*, input_proj.*,
    output_proj.*, center_feature_scale_proj_*), so InternImage checkpoints load unchanged."""

    def __init__(self, channels=64, kernel_size=3, dw_kernel_size=None, stride=1, pad=1, dilation=1, group=4,
                 offset_scale=1.0, act_layer="GELU", norm_layer="LN", center_feature_scale=False):
        super().__init__()
        if channels % group != 0:
            raise ValueError(f"channels must be divisible by group, but got {channels} and {group}")
        d = channels // group
        dw_kernel_size = dw_kernel_size if dw_kernel_size is not None else kernel_size
        if not ((d & (d - 1) == 0) and d != 0):
            warnings.warn("You'd better set channels in DCNv3 to make the dimension of each attention head a power of 2 "
                          "which is more efficient in our HIP implementation (16-byte lane gathers).")
        self.offset_scale, self.channels, self.kernel_size, self.dw_kernel_size = offset_scale, channels, kernel_size, dw_kernel_size
        self.stride, self.dilation, self.pad, self.group, self.group_channels = stride, dilation, pad, group, d
        self.center_feature_scale = center_feature_scale
        self.dw_conv = nn.Sequential(
            nn.Conv2d(channels, channels, kernel_size=dw_kernel_size, stride=1, padding=(dw_kernel_size - 1) // 2,
                      groups=channels),
            build_norm_layer(channels, norm_layer, "channels_first", "channels_last"),
            build_act_layer(act_layer))
        self.offset = nn.Linear(channels, group * kernel_size * kernel_size * 2)
        self.mask = nn.Linear(channels, group * kernel_size * kernel_size)
        self.input_proj = nn.Linear(channels, channels)
        self.output_proj = nn.Linear(channels, channels)
        self._reset_parameters()
        if center_feature_scale:
            self.center_feature_scale_proj_weight = nn.Parameter(torch.zeros((group, channels), dtype=torch.float))
            self.center_feature_scale_proj_bias = nn.Parameter(torch.zeros((group,), dtype=torch.float))

    def _reset_parameters(self):
        constant_(self.offset.weight.data, 0.0)
        constant_(self.offset.bias.data, 0.0)
        constant_(self.mask.weight.data, 0.0)
        constant_(self.mask.bias.data, 0.0)
        xavier_uniform_(self.input_proj.weight.data)
        constant_(self.input_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, input):
        """(N, H, W, C) -> (N, H, W, C)"""
        N, H, W, _ = input.shape
        x = self.input_proj(input)
        x_proj = x
        dtype = x.dtype
        x1 = self.dw_conv(input.permute(0, 3, 1, 2))
        offset = self.offset(x1)
        mask = F.softmax(self.mask(x1).reshape(N, H, W, self.group, -1), -1).reshape(N, H, W, -1).type(dtype)
        x = DCNv3Function.apply(x.to(torch.float32), offset.to(torch.float32), mask.to(torch.float32), self.kernel_size,
                                self.kernel_size, self.stride, self.stride, self.pad, self.pad, self.dilation,
                                self.dilation, self.group, self.group_channels, self.offset_scale, 256).to(dtype)
        if self.center_feature_scale:
            cfs = F.linear(x1, self.center_feature_scale_proj_weight.to(x1.dtype),
                           self.center_feature_scale_proj_bias.to(x1.dtype)).sigmoid()
            cfs = cfs[..., None].repeat(1, 1, 1, 1, self.channels // self.group).flatten(-2)
            x = x * (1 - cfs) + x_proj * cfs
        return self.output_proj(x)
