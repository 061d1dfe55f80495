"""Multi-scale deformable attention: host-side mirror of the reference operator + module interfaces.

Boundary B3 of SURVEY.md section 8b.  Same names, argument meaning and error behaviour as

* ``MultiScaleDeformableAttention.ms_deform_attn_forward / ms_deform_attn_backward`` -- the pybind module of
  visionllmv2/model/unipose/ops/src/vision.cpp:13-16 (signatures ms_deform_attn.h:20-61) and the object returned
  by HF's ``load_cuda_kernels()`` (modeling_ov_grounding_dino_mask_dn.py:110, 147, 170);
* ``MSDeformAttnFunction``   -- visionllmv2/model/unipose/ops/functions/ms_deform_attn_func.py:21-38
  (== ``MultiScaleDeformableAttentionFunction``, modeling_ov_grounding_dino_mask_dn.py:134-180);
* ``MultiScaleDeformableAttnFunction`` (mmcv flavour: fp32 cast, in-place grads) -- mmcv/mmcv/ops/multi_scale_deform_attn.py:23-97;
* ``MSDeformAttn``           -- visionllmv2/model/unipose/ops/modules/ms_deform_attn.py:34-145;
* ``MultiScaleDeformableAttention`` (mmcv module) -- mmcv/mmcv/ops/multi_scale_deform_attn.py:162-367;
* ``GroundingDinoMultiscaleDeformableAttention`` -- modeling_ov_grounding_dino_mask_dn.py:646-784.

The compute is libvllm_hip.so (visionllm_amd/csrc/msda.hip).  Like the reference's native extension
("Not implement on cpu", src/cpu/ms_deform_attn_cpu.cpp:16-40) the op raises for CPU tensors; unlike the
reference modules we do NOT silently fall back to a pure-torch path (modeling_ov_grounding_dino_mask_dn.py:777-779
would mask a broken kernel).
"""
import ctypes
import math
import warnings
import weakref

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


def _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    # mirrors the AT_ASSERTM block of ms_deform_attn_cuda.cu:28-52
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor (ms_deform_attn: Not implemented on the CPU)")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("ms_deform_attn: value [B,S,M,D], sampling_loc [B,Lq,M,L,P,2], attn_weight [B,Lq,M,L,P]")
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_loc.shape
    if M2 != M or two != 2 or tuple(attn_weight.shape) != (B, Lq, M, L, P) or sampling_loc.shape[0] != B:
        raise RuntimeError("ms_deform_attn: inconsistent shapes")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("ms_deform_attn: spatial_shapes / level_start_index must be int64")
    if tuple(spatial_shapes.shape) != (L, 2) or level_start_index.numel() != L:
        raise RuntimeError("ms_deform_attn: spatial_shapes must be [L,2] and level_start_index [L]")
    step = min(B, int(im2col_step)) if B > 0 else 1
    if step <= 0 or (B > 0 and B % step != 0):
        raise RuntimeError(f"batch({B}) must divide im2col_step({step})")
    return B, S, M, D, L, Lq, P


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """-> Tensor[B, Lq, M*D].  float32 / float64 (as the reference's AT_DISPATCH_FLOATING_TYPES) and, as an
    extension, bfloat16 value with float32 locations/weights."""
    B, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                         im2col_step)
    lib = _lib.lib()
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        st = _lib.current_stream(value.device)
        args = (_lib.ptr(value), _lib.ptr(spatial_shapes), _lib.ptr(level_start_index), _lib.ptr(sampling_loc),
                _lib.ptr(attn_weight), B, S, M, D, L, Lq, P, _lib.ptr(out), st)
        if value.dtype == torch.float32:
            if sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32:
                raise RuntimeError("ms_deform_attn_forward: all floating inputs must share the value dtype (float32)")
            geo = known_geometry(spatial_shapes, Lq)   # no synchronisation: GEO_UNKNOWN unless a module / caller looked already
            _lib.check(lib.vllm_msda_forward_f32_geo(*args[:12], geo, *args[12:]), "vllm_msda_forward_f32")
        elif value.dtype == torch.float64:
            if sampling_loc.dtype != torch.float64 or attn_weight.dtype != torch.float64:
                raise RuntimeError("ms_deform_attn_forward: all floating inputs must share the value dtype (float64)")
            _lib.check(lib.vllm_msda_forward_f64(*args), "vllm_msda_forward_f64")
        elif value.dtype == torch.bfloat16:
            if sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32:
                raise RuntimeError("ms_deform_attn_forward(bf16 value): sampling_loc / attn_weight must be float32")
            _lib.check(lib.vllm_msda_forward_bf16(*args), "vllm_msda_forward_bf16")
        else:
            raise RuntimeError(f"ms_deform_attn_forward: unsupported dtype {value.dtype}")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step=64):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (list, as ms_deform_attn.h:41-61)."""
    B, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                         im2col_step)
    if not grad_output.is_cuda:
        raise RuntimeError("grad_output must be a CUDA tensor")
    grad_output = grad_output.contiguous()
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"ms_deform_attn_backward: unsupported dtype {value.dtype}")
    gv = torch.zeros_like(value)
    # (the reference zero-fills all three, ms_deform_attn_cuda.cu:118-120; the kernel of the encoder shape writes every element of the two
    #  per-point gradients itself: no memset for them)
    gl = torch.empty_like(sampling_loc)
    gw = torch.empty_like(attn_weight)
    if not (value.dtype == torch.float32 and value.is_contiguous() and sampling_loc.is_contiguous() and attn_weight.is_contiguous() and
            _lib.lib().vllm_msda_backward_f32_writes_point_grads(_lib.ptr(value), _lib.ptr(sampling_loc), _lib.ptr(grad_output), _lib.ptr(gv),
                                                                 _lib.ptr(gl), B, S, M, D, L, Lq, P)):
        gl.zero_()
        gw.zero_()
    _backward_into(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, gv, gl, gw,
                   (B, S, M, D, L, Lq, P))
    return [gv, gl, gw]


def _backward_into(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, gv, gl, gw, dims):
    B, S, M, D, L, Lq, P = dims
    lib = _lib.lib()
    fn = lib.vllm_msda_backward_f32 if value.dtype == torch.float32 else lib.vllm_msda_backward_f64
    with torch.cuda.device(value.device):
        _lib.check(fn(_lib.ptr(value), _lib.ptr(spatial_shapes), _lib.ptr(level_start_index), _lib.ptr(sampling_loc),
                      _lib.ptr(attn_weight), _lib.ptr(grad_output), B, S, M, D, L, Lq, P, _lib.ptr(gv), _lib.ptr(gl),
                      _lib.ptr(gw), _lib.current_stream(value.device)), "vllm_msda_backward")


def ms_deform_attn_backward_(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                             grad_value, grad_sampling_loc, grad_attn_weight, im2col_step=64):
    """mmcv ``_ext`` flavour: accumulates into caller-provided (zero-filled) gradients
    (mmcv/ops/csrc/pytorch/ms_deform_attn.cpp:48-60)."""
    dims = _check_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    _backward_into(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output.contiguous(),
                   grad_value, grad_sampling_loc, grad_attn_weight, dims)


def sample_index(spatial_shapes, sampling_loc):
    """Parity tooling: integer part of the sampling (h_low, w_low, mask) computed by the kernels' own device
    function.  mask bit0 = point accepted, bits 1-4 = corner 1..4 in bounds."""
    B, Lq, M, L, P, _ = sampling_loc.shape
    dev = sampling_loc.device
    h = torch.empty((B, Lq, M, L, P), dtype=torch.int32, device=dev)
    w = torch.empty_like(h)
    mk = torch.empty((B, Lq, M, L, P), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vllm_msda_sample_index_f32(
            _lib.ptr(spatial_shapes), _lib.ptr(sampling_loc.contiguous().float()), B, M, L, Lq, P, _lib.ptr(h),
            _lib.ptr(w), _lib.ptr(mk), _lib.current_stream(dev)), "vllm_msda_sample_index_f32")
    return h, w, mk


class MSDeformAttnFunction(Function):
    """ms_deform_attn_func.py:21-38 / modeling_ov_grounding_dino_mask_dn.py:134-180."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attw = ctx.saved_tensors
        gv, gl, gw = ms_deform_attn_backward(value, shapes, lsi, loc, attw, grad_output, ctx.im2col_step)
        return gv, None, None, gl, gw, None


MultiScaleDeformableAttentionFunction = MSDeformAttnFunction  # HF name


class MultiScaleDeformableAttnFunction(Function):
    """mmcv flavour (multi_scale_deform_attn.py:23-97): inputs are cast to float32 (``custom_fwd``)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        if value.dtype != torch.float64:
            value, sampling_locations, attention_weights = (value.float(), sampling_locations.float(),
                                                             attention_weights.float())
        output = ms_deform_attn_forward(value.contiguous(), value_spatial_shapes, value_level_start_index,
                                        sampling_locations.contiguous(), attention_weights.contiguous(), im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attw = ctx.saved_tensors
        # (the allocating form: zero-fills only what the kernel that runs does not write itself)
        gv, gl, gw = ms_deform_attn_backward(value.contiguous(), shapes, lsi, loc.contiguous(), attw.contiguous(),
                                             grad_output.to(value.dtype).contiguous(), ctx.im2col_step)
        dv, dl, dw = ctx.in_dtypes
        return gv.to(dv), None, None, gl.to(dl), gw.to(dw), None


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


def _init_msda_parameters(mod, n_heads, n_levels, n_points):
    """Shared initialisation (ms_deform_attn.py:66-81 == multi_scale_deform_attn.py:238-257)."""
    nn.init.constant_(mod.sampling_offsets.weight.data, 0.0)
    thetas = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(n_heads, 1, 1, 2).repeat(
        1, n_levels, n_points, 1)
    for i in range(n_points):
        grid_init[:, :, i, :] *= i + 1
    with torch.no_grad():
        mod.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
    nn.init.constant_(mod.attention_weights.weight.data, 0.0)
    nn.init.constant_(mod.attention_weights.bias.data, 0.0)
    nn.init.xavier_uniform_(mod.value_proj.weight.data)
    nn.init.constant_(mod.value_proj.bias.data, 0.0)
    nn.init.xavier_uniform_(mod.output_proj.weight.data)
    nn.init.constant_(mod.output_proj.bias.data, 0.0)


def _sampling_locations(reference_points, sampling_offsets, spatial_shapes, n_points, use_4d_normalizer=False):
    if reference_points.shape[-1] == 2:
        offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        return reference_points[:, :, None, :, None, :] + \
            sampling_offsets / offset_normalizer[None, None, None, :, None, :]
    if reference_points.shape[-1] == 4:
        if use_4d_normalizer:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            return reference_points[:, :, None, :, None, :2] + \
                sampling_offsets / offset_normalizer[None, None, None, :, None, :] * \
                reference_points[:, :, None, :, None, 2:] * 0.5
        return reference_points[:, :, None, :, None, :2] + \
            sampling_offsets / n_points * reference_points[:, :, None, :, None, 2:] * 0.5
    raise ValueError(
        "Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))


def msda_layer_fused_ok(query, input_flatten, *linears, reference_points=None):
    """The fused HIP layer serves inference-dtype (bf16) modules on the GPU; anything else keeps the composed path."""
    if not (query.is_cuda and query.dtype == torch.bfloat16 and input_flatten.dtype == torch.bfloat16):
        return False
    if torch.is_grad_enabled():
        # training differentiates through the composed path (autograd Function around the operator): trainable
        # parameters, or a frozen module whose INPUTS carry gradients to upstream layers (the fused call has no grad_fn)
        if any(p.requires_grad for lin in linears for p in lin.parameters()):
            return False
        if query.requires_grad or input_flatten.requires_grad or (reference_points is not None and reference_points.requires_grad):
            return False
    d_model = query.shape[-1]
    return d_model % 64 == 0 and all(lin.weight.dtype == torch.bfloat16 and lin.bias is not None for lin in linears)


GEO_UNKNOWN, GEO_PYRAMID, GEO_GENERAL, GEO_NESTED = 0, 1, 2, 3   # VLLM_GEO_* of include/vllm_hip.h
_SHAPE_FACTS = {}   # id(spatial_shapes) -> (weakref, tensor version, sum of H * W, geometry)


def _tensor_version(t):
    """Autograd version counter, or None for tensors that have none (created under torch.inference_mode())."""
    try:
        return t._version
    except RuntimeError:
        return None


def shape_facts(spatial_shapes):
    """(sum_l H_l * W_l, geometry) of a [L, 2] shape tensor, as Python ints.

    The sum is the quantity the reference's modules compare with the value length on every call (ms_deform_attn.py:100,
    multi_scale_deform_attn.py:319, ...mask_dn.py:741) -- a host synchronisation per layer.  The same read-back tells
    whether the level maps are an exact 2x pyramid (GEO_PYRAMID), its ceil / floor-divided variants (GEO_NESTED) or neither (GEO_GENERAL), which lets the native operator
    enqueue ONE kernel instead of two (vllm_msda_forward_f32_geo).  The det heads pass the SAME tensor object to every
    encoder / decoder layer, so the result is remembered per tensor object and autograd version (an in-place change bumps
    the version; a dead object's id can be reused, hence the weak reference): one synchronisation per forward pass instead
    of one per layer, same check.  Tensors without a version counter (inference tensors) are read back on every call: an
    in-place change could not be noticed."""
    key = id(spatial_shapes)
    ver = _tensor_version(spatial_shapes)
    ent = _SHAPE_FACTS.get(key)
    if ver is not None and ent is not None and ent[0]() is spatial_shapes and ent[1] == ver:
        return ent[2], ent[3]
    hw = spatial_shapes.detach().reshape(-1, 2).tolist()    # ONE device -> host copy
    total = sum(int(h) * int(w) for h, w in hw)
    dims = [(int(h), int(w)) for h, w in hw]
    exact = nested_maps(dims) and all((h << l) == dims[0][0] and (w << l) == dims[0][1] for l, (h, w) in enumerate(dims))
    geo = GEO_PYRAMID if exact else GEO_NESTED if nested_maps(dims) else GEO_GENERAL
    if ver is not None:
        if len(_SHAPE_FACTS) >= 64:
            _SHAPE_FACTS.clear()
        _SHAPE_FACTS[key] = (weakref.ref(spatial_shapes), ver, total, geo)
    return total, geo


def nested_maps(hw):
    """The pyramid-item kernel's predicate (csrc/msda_sample.hpp, geometry_is_nested): 1-4 levels, each the previous one
    halved, rounded either way -- exact 2x pyramids and the ceil-divided maps of a detection backbone (100x167, 50x84, ...)."""
    if not 1 <= len(hw) <= 4 or hw[0][0] <= 0 or hw[0][1] <= 0 or hw[0][0] > 16384 or hw[0][1] > 16384:   # (16-bit packed corner coordinates)
        return False
    nty, ntx = (hw[0][0] + 7) >> 3, (hw[0][1] + 15) >> 4
    for l in range(1, len(hw)):
        (hp, wp), (h, w) = hw[l - 1], hw[l]
        if not (h > 0 and w > 0 and hp >> 1 <= h <= (hp + 1) >> 1 and wp >> 1 <= w <= (wp + 1) >> 1 and
                h <= nty * (8 >> l) and w <= ntx * (16 >> l)):
            return False
    return True


def level_pixels(spatial_shapes):
    """sum_l H_l * W_l as a Python int (see shape_facts)."""
    return shape_facts(spatial_shapes)[0]


def known_geometry(spatial_shapes, num_queries):
    """GEO_* for the native operator if it is already known for this tensor object WITHOUT a synchronisation (a module's
    `level_pixels` check, or an earlier call of `remember_geometry`), else GEO_UNKNOWN.  The pyramid kernel serves the
    encoder case only: the queries must be the cells of the maps."""
    ver = _tensor_version(spatial_shapes)
    ent = _SHAPE_FACTS.get(id(spatial_shapes))
    if ver is None or ent is None or ent[0]() is not spatial_shapes or ent[1] != ver:
        return GEO_UNKNOWN
    if ent[3] in (GEO_PYRAMID, GEO_NESTED) and ent[2] != num_queries:
        return GEO_GENERAL
    return ent[3]


def remember_geometry(spatial_shapes):
    """Optional, for callers of the bare operator (ms_deform_attn_forward) outside the module mirrors: one read-back now,
    one kernel launch per call afterwards."""
    return shape_facts(spatial_shapes)[1]


def msda_layer_forward(query, reference_points, input_flatten, spatial_shapes, level_start_index, padding_mask,
                       value_proj, sampling_offsets, attention_weights, output_proj, n_heads, n_levels, n_points,
                       use_4d_normalizer=False):
    """value_proj -> offsets / weights linears -> softmax -> locations -> operator -> output_proj in ONE C call
    (vllm_msda_layer_forward; replaces ms_deform_attn.py:102-145 / ...mask_dn.py:729-782 for bf16 modules).
    query [B, Lq, C] bf16 (position embedding already added), reference_points [B, Lq, L, 2|4], input_flatten [B, S, C]."""
    return msda_layer_prepare(query, reference_points, input_flatten, spatial_shapes, level_start_index, padding_mask,
                              value_proj, sampling_offsets, attention_weights, output_proj, n_heads, n_levels, n_points,
                              use_4d_normalizer)()


def msda_layer_prepare(query, reference_points, input_flatten, spatial_shapes, level_start_index, padding_mask,
                       value_proj, sampling_offsets, attention_weights, output_proj, n_heads, n_levels, n_points,
                       use_4d_normalizer=False):
    """Everything of msda_layer_forward that needs no device result -- argument checks, the descriptor, contiguous operands,
    workspace, output -- and a zero-argument callable that enqueues the C call.  The module mirrors prepare FIRST and only
    then evaluate the reference's `assert (spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum() == Len_in` (a host
    synchronisation): the host-side preparation then overlaps the GPU's backlog instead of sitting between two layers."""
    B, Lq, C = query.shape
    S = input_flatten.shape[1]
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))
    # the native kernels index these tensors by (B, Lq, n_levels): a broadcastable view must be materialised here
    if reference_points.dim() != 4 or reference_points.shape[0] not in (1, B) or reference_points.shape[1] not in (1, Lq) or \
            reference_points.shape[2] not in (1, n_levels):
        raise RuntimeError(f"msda_layer_forward: reference_points {tuple(reference_points.shape)} does not broadcast to "
                           f"({B}, {Lq}, {n_levels}, 2|4)")
    reference_points = reference_points.expand(B, Lq, n_levels, reference_points.shape[-1])
    if tuple(input_flatten.shape) != (B, S, C):
        raise RuntimeError(f"msda_layer_forward: input_flatten {tuple(input_flatten.shape)} vs query {tuple(query.shape)}")
    if tuple(spatial_shapes.shape) != (n_levels, 2) or level_start_index.numel() != n_levels:
        raise RuntimeError(f"msda_layer_forward: spatial_shapes {tuple(spatial_shapes.shape)} / level_start_index "
                           f"({level_start_index.numel()}) do not describe {n_levels} levels")
    if padding_mask is not None and tuple(padding_mask.shape) != (B, S):
        raise RuntimeError(f"msda_layer_forward: padding mask {tuple(padding_mask.shape)} must be ({B}, {S})")
    L = _lib.lib()
    desc = _lib.VllmMsdaLayerDesc()
    desc.d_model, desc.n_heads, desc.n_levels, desc.n_points = C, n_heads, n_levels, n_points
    desc.ref_dim, desc.use_4d_normalizer = reference_points.shape[-1], int(bool(use_4d_normalizer))
    keep = []
    for name, lin in (("value_proj", value_proj), ("sampling_offsets", sampling_offsets),
                      ("attention_weights", attention_weights), ("output_proj", output_proj)):
        w, b = lin.weight.detach().contiguous(), lin.bias.detach().to(torch.bfloat16).contiguous()
        keep += [w, b]
        setattr(desc, name + "_w", w.data_ptr())
        setattr(desc, name + "_b", b.data_ptr())
    q = query.contiguous()
    x = input_flatten.contiguous()
    ref = reference_points.to(torch.float32).contiguous()
    shapes = spatial_shapes.to(device=q.device, dtype=torch.int64).contiguous()
    lsi = level_start_index.to(device=q.device, dtype=torch.int64).contiguous()
    mask = None if padding_mask is None else padding_mask.to(torch.uint8).contiguous()
    nbytes = L.vllm_msda_layer_workspace_bytes(ctypes.byref(desc), B, Lq, S)
    if nbytes < 0:
        _lib.check(-1, "vllm_msda_layer_workspace_bytes")
    ws = _lib.workspace(q.device, nbytes)
    out = torch.empty_like(q)

    def launch():
        # by now the module mirror has run the reference's `(H * W).sum() == Len_in` check on this tensor object, so the
        # geometry is known without another synchronisation (GEO_UNKNOWN otherwise: the device decides, two launches more)
        desc.geometry = known_geometry(spatial_shapes, Lq)
        with torch.cuda.device(q.device):
            _lib.check(L.vllm_msda_layer_forward(ctypes.byref(desc), _lib.ptr(q), _lib.ptr(ref), _lib.ptr(x), _lib.ptr(mask),
                                                 _lib.ptr(shapes), _lib.ptr(lsi), B, Lq, S, _lib.ptr(out), _lib.ptr(ws),
                                                 ws.numel(), _lib.current_stream(q.device)), "vllm_msda_layer_forward")
        launch.keep = keep   # (parameters referenced by the descriptor stay alive until the call has been enqueued)
        return out
    return launch


def _msda_apply_fp32(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step):
    """The reference upcasts to fp32 around the op (ms_deform_attn.py:131-139; ...mask_dn.py:764-766)."""
    dtype = value.dtype
    if dtype not in (torch.float32, torch.float64):
        out = MSDeformAttnFunction.apply(value.to(torch.float32).contiguous(), spatial_shapes, level_start_index,
                                         sampling_locations.to(torch.float32).contiguous(),
                                         attention_weights.to(torch.float32).contiguous(), im2col_step)
        return out.to(dtype)
    return MSDeformAttnFunction.apply(value.contiguous(), spatial_shapes, level_start_index,
                                      sampling_locations.to(dtype).contiguous(),
                                      attention_weights.to(dtype).contiguous(), im2col_step)


class MSDeformAttn(nn.Module):
    """UniPose / Deformable-DETR module (unipose/ops/modules/ms_deform_attn.py:34-145)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, use_4D_normalizer=False):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("You'd better set d_model in MSDeformAttn to make the dimension of each attention head a "
                          "power of 2 which is more efficient in our HIP implementation (16-byte lane gathers).")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self.use_4D_normalizer = use_4D_normalizer
        self._reset_parameters()

    def _reset_parameters(self):
        _init_msda_parameters(self, self.n_heads, self.n_levels, self.n_points)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        fused = None
        if msda_layer_fused_ok(query, input_flatten, self.value_proj, self.sampling_offsets, self.attention_weights,
                               self.output_proj, reference_points=reference_points):   # bf16 inference: one native call
            fused = msda_layer_prepare(query, reference_points, input_flatten, input_spatial_shapes,
                                       input_level_start_index, input_padding_mask, self.value_proj,
                                       self.sampling_offsets, self.attention_weights, self.output_proj, self.n_heads,
                                       self.n_levels, self.n_points, self.use_4D_normalizer)
        assert level_pixels(input_spatial_shapes) == Len_in
        if fused is not None:
            return fused()
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        if self.sampling_offsets.bias.dtype != query.dtype:
            self.sampling_offsets.bias.data = self.sampling_offsets.bias.data.to(query.dtype)
        sampling_offsets = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(N, Len_q, self.n_heads, self.n_levels,
                                                                  self.n_points)
        sampling_locations = _sampling_locations(reference_points, sampling_offsets, input_spatial_shapes,
                                                 self.n_points, self.use_4D_normalizer)
        output = _msda_apply_fp32(value, input_spatial_shapes, input_level_start_index, sampling_locations,
                                  attention_weights, self.im2col_step)
        return self.output_proj(output)


class MSDeformAttnKeyAware(MSDeformAttn):
    """The "key-aware" signature variant (unipose/ops/modules/ms_deform_attn_key_aware.py:33-132): same parameters, same arithmetic,
    ``forward(query, key, reference_points, ...)`` with a ``key`` of shape (N, 1, C) that the reference's forward never reads (:83-132).
    Not imported by modeling_unipose.py; provided so that ``from ...ms_deform_attn_key_aware import MSDeformAttn`` has a drop-in
    (``visionllm_amd.compat`` users alias it)."""

    def forward(self, query, key, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        del key   # (unused by the reference as well)
        return super().forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                               input_padding_mask)


class MultiScaleDeformableAttention(nn.Module):
    """mmcv module (mmcv/ops/multi_scale_deform_attn.py:162-367): (num_query, bs, C) unless batch_first, residual
    ``identity`` and dropout on the output."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        if not _is_power_of_2(embed_dims // num_heads):
            warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the dimension of each "
                          "attention head a power of 2 which is more efficient in our HIP implementation.")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        _init_msda_parameters(self, self.num_heads, self.num_levels, self.num_points)
        self._is_init = True

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        fused = None
        if msda_layer_fused_ok(query, value, self.value_proj, self.sampling_offsets, self.attention_weights,
                               self.output_proj, reference_points=reference_points):
            fused = msda_layer_prepare(query, reference_points, value, spatial_shapes, level_start_index,
                                       key_padding_mask, self.value_proj, self.sampling_offsets,
                                       self.attention_weights, self.output_proj, self.num_heads, self.num_levels,
                                       self.num_points)
        assert level_pixels(spatial_shapes) == num_value
        if fused is not None:
            output = fused()
            if not self.batch_first:
                output = output.permute(1, 0, 2)
            return self.dropout(output) + identity
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(bs, num_query, self.num_heads, self.num_levels,
                                                             self.num_points, 2)
        attention_weights = self.attention_weights(query).view(bs, num_query, self.num_heads,
                                                               self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(bs, num_query, self.num_heads, self.num_levels,
                                                               self.num_points)
        sampling_locations = _sampling_locations(reference_points, sampling_offsets, spatial_shapes, self.num_points)
        output = self._operator(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return self.dropout(output) + identity

    def _operator(self, value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
        # multi_scale_deform_attn.py:26 (custom_fwd(cast_inputs=torch.float32)): the operator runs in fp32 whatever the module's dtype
        return MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                                      attention_weights, self.im2col_step).to(value.dtype)


class MultiScaleDeformableAttentionOptimized(MultiScaleDeformableAttention):
    """The fork-added mmcv class (mmcv/ops/multi_scale_deform_attn_optimized.py:160-364, exported at ops/__init__.py:42): the same
    module -- same parameters, same forward -- on the ``MultiScaleDeformableAttention`` extension's ABI (:18-21): no forced fp32 cast
    (``custom_fwd`` without ``cast_inputs``, :27), sampling locations / attention weights follow value's dtype (:55-56), the gradients
    are the extension's return value (:84-91).  fp32 / fp64 values run in their own dtype; for bf16 / fp16 values (which the
    reference's extension does not dispatch: AT_DISPATCH_FLOATING_TYPES) the operator is evaluated in fp32, as everywhere else here."""

    def _operator(self, value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
        return _msda_apply_fp32(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, self.im2col_step)


class GroundingDinoMultiscaleDeformableAttention(nn.Module):
    """HF / Grounding-DINO module (modeling_ov_grounding_dino_mask_dn.py:646-784).  ``config`` needs ``d_model``,
    ``num_feature_levels`` and (ignored here) ``disable_custom_kernels``."""

    def __init__(self, config, num_heads: int, n_points: int):
        super().__init__()
        if config.d_model % num_heads != 0:
            raise ValueError(
                f"embed_dim (d_model) must be divisible by num_heads, but got {config.d_model} and {num_heads}")
        dim_per_head = config.d_model // num_heads
        if not ((dim_per_head & (dim_per_head - 1) == 0) and dim_per_head != 0):
            warnings.warn("You'd better set embed_dim (d_model) in GroundingDinoMultiscaleDeformableAttention to make "
                          "the dimension of each attention head a power of 2 (16-byte lane gathers in the HIP kernel).")
        self.im2col_step = 64
        self.d_model = config.d_model
        self.n_levels = config.num_feature_levels
        self.n_heads = num_heads
        self.n_points = n_points
        self.sampling_offsets = nn.Linear(config.d_model, num_heads * self.n_levels * n_points * 2)
        self.attention_weights = nn.Linear(config.d_model, num_heads * self.n_levels * n_points)
        self.value_proj = nn.Linear(config.d_model, config.d_model)
        self.output_proj = nn.Linear(config.d_model, config.d_model)
        self.disable_custom_kernels = getattr(config, "disable_custom_kernels", False)
        self._reset_parameters()

    def _reset_parameters(self):
        _init_msda_parameters(self, self.n_heads, self.n_levels, self.n_points)

    def with_pos_embed(self, tensor, position_embeddings):
        return tensor if position_embeddings is None else tensor + position_embeddings

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                position_embeddings=None, reference_points=None, spatial_shapes=None, level_start_index=None,
                output_attentions: bool = False):
        if position_embeddings is not None:
            hidden_states = self.with_pos_embed(hidden_states, position_embeddings)
        batch_size, num_queries, _ = hidden_states.shape
        batch_size, sequence_length, _ = encoder_hidden_states.shape
        fused = None
        if not output_attentions and msda_layer_fused_ok(hidden_states, encoder_hidden_states, self.value_proj,
                                                         self.sampling_offsets, self.attention_weights,
                                                         self.output_proj, reference_points=reference_points):
            # (the attention weights stay inside the fused call; ask for output_attentions to get them)
            fused = msda_layer_prepare(hidden_states, reference_points, encoder_hidden_states, spatial_shapes,
                                       level_start_index, None if attention_mask is None else ~attention_mask,
                                       self.value_proj, self.sampling_offsets, self.attention_weights,
                                       self.output_proj, self.n_heads, self.n_levels, self.n_points)
        if level_pixels(spatial_shapes) != sequence_length:
            raise ValueError(
                "Make sure to align the spatial shapes with the sequence length of the encoder hidden states")
        if fused is not None:
            return fused(), None
        value = self.value_proj(encoder_hidden_states)
        if attention_mask is not None:
            value = value.masked_fill(~attention_mask[..., None], float(0))
        value = value.view(batch_size, sequence_length, self.n_heads, self.d_model // self.n_heads)
        if self.sampling_offsets.bias.dtype != hidden_states.dtype:
            self.sampling_offsets.bias.data = self.sampling_offsets.bias.data.to(hidden_states.dtype)
        sampling_offsets = self.sampling_offsets(hidden_states).view(batch_size, num_queries, self.n_heads,
                                                                      self.n_levels, self.n_points, 2)
        attention_weights = self.attention_weights(hidden_states).view(batch_size, num_queries, self.n_heads,
                                                                        self.n_levels * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(batch_size, num_queries, self.n_heads,
                                                                  self.n_levels, self.n_points)
        sampling_locations = _sampling_locations(reference_points, sampling_offsets, spatial_shapes, self.n_points)
        output = _msda_apply_fp32(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                                  self.im2col_step)
        output = output.to(self.output_proj.weight.dtype)
        output = self.output_proj(output)
        return output, attention_weights
