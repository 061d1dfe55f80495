"""ORACLE -- test infrastructure only.

numpy/ctypes front-end to ``oracle/msda_oracle.c`` plus a torch restatement of the
reference's pure-PyTorch MSDA twin (the path the reference itself runs on CPU):

* ``forward`` / ``backward`` / ``sample_index``: C restatement of the native kernel
  (visionllmv2/model/unipose/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84, 237-298, 87-161).
* ``grid_sample_twin``: restatement of ``multi_scale_deformable_attn_pytorch``
  (mmcv/mmcv/ops/multi_scale_deform_attn.py:100-159 ==
  visionllmv2/model/unipose/ops/functions/ms_deform_attn_func.py:41-61 ==
  visionllmv2/model/grounding_dino/modeling_ov_grounding_dino_mask_dn.py:607-643).
  Used as the timed "reference CPU path" in bench.py's cpu_baseline.
"""
import ctypes

import numpy as np

from . import build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        assert _lib.msda_oracle_abi_version() == 1
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attw):
    dt = np.float64 if value.dtype == np.float64 else np.float32
    value = np.ascontiguousarray(value, dtype=dt)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attw = np.ascontiguousarray(attw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and lsi.shape == (L,)
    assert attw.shape == (B, Lq, M, L, P)
    return dt, value, shapes, lsi, loc, attw, (B, S, M, D, L, Lq, P)


def forward(value, shapes, lsi, loc, attw):
    """value [B,S,M,D], shapes [L,2] (H,W), lsi [L], loc [B,Lq,M,L,P,2] (x,y), attw [B,Lq,M,L,P] -> [B,Lq,M*D]."""
    dt, value, shapes, lsi, loc, attw, (B, S, M, D, L, Lq, P) = _prep(value, shapes, lsi, loc, attw)
    out = np.empty((B, Lq, M * D), dtype=dt)
    fn = lib().msda_oracle_forward_f64 if dt == np.float64 else lib().msda_oracle_forward_f32
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attw), B, S, M, D, L, Lq, P, _p(out))
    return out


def backward(value, shapes, lsi, loc, attw, grad_out):
    dt, value, shapes, lsi, loc, attw, (B, S, M, D, L, Lq, P) = _prep(value, shapes, lsi, loc, attw)
    grad_out = np.ascontiguousarray(grad_out, dtype=dt).reshape(B, Lq, M * D)
    gv = np.zeros_like(value)
    gl = np.zeros_like(loc)
    gw = np.zeros_like(attw)
    fn = lib().msda_oracle_backward_f64 if dt == np.float64 else lib().msda_oracle_backward_f32
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(attw), _p(grad_out),
       B, S, M, D, L, Lq, P, _p(gv), _p(gl), _p(gw))
    return gv, gl, gw


def sample_index(shapes, loc):
    """Integer part of the sampling: (h_low, w_low, mask) per point; mask bit0 = accepted, bits1-4 = corners."""
    dt = np.float64 if loc.dtype == np.float64 else np.float32
    loc = np.ascontiguousarray(loc, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    B, Lq, M, L, P, _ = loc.shape
    h = np.empty((B, Lq, M, L, P), dtype=np.int32)
    w = np.empty_like(h)
    mk = np.empty((B, Lq, M, L, P), dtype=np.uint8)
    fn = lib().msda_oracle_sample_index_f64 if dt == np.float64 else lib().msda_oracle_sample_index_f32
    fn(_p(shapes), _p(loc), B, M, L, Lq, P, _p(h), _p(w), _p(mk))
    return h, w, mk


def grid_sample_twin(value, value_spatial_shapes, sampling_locations, attention_weights):
    """torch restatement of the reference's pure-PyTorch twin (see module docstring)."""
    import torch
    import torch.nn.functional as F

    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, _, num_levels, num_points, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    value_list = value.split(sizes, dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(value_spatial_shapes):
        h, w = int(h), int(w)
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries, num_levels * num_points)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, num_heads * embed_dims, num_queries)
    return out.transpose(1, 2).contiguous()


def level_start_index(shapes):
    shapes = np.asarray(shapes, dtype=np.int64)
    areas = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(areas)[:-1]]).astype(np.int64)


def layer_forward(query, reference_points, input_flatten, shapes, lsi, padding_mask, params, n_heads, n_levels, n_points,
                  use_4d_normalizer=False, dtype=np.float64):
    """CPU restatement of the deformable-attention LAYER (test infrastructure).

    Follows MSDeformAttn.forward, visionllmv2/model/unipose/ops/modules/ms_deform_attn.py:83-145 (== mmcv
    multi_scale_deform_attn.py:318-367 == ...mask_dn.py:729-782): value_proj (:106), key-padding zero fill (:107-108),
    sampling_offsets / attention_weights linears (:110-111), softmax over L*P (:112), locations for 2-d (:114-117) and
    4-d reference points (:118-126), the operator (C oracle, fp64 when dtype is float64), output_proj (:144).
    ``params`` maps value_proj / sampling_offsets / attention_weights / output_proj to (weight [out, in], bias)."""
    q = np.asarray(query, dtype)
    x = np.asarray(input_flatten, dtype)
    ref = np.asarray(reference_points, dtype)
    B, Lq, C = q.shape
    S = x.shape[1]
    M, L, P = n_heads, n_levels, n_points

    def lin(name, t):
        w, b = params[name]
        return t @ np.asarray(w, dtype).T + np.asarray(b, dtype)

    value = lin("value_proj", x)
    if padding_mask is not None:
        value = np.where(np.asarray(padding_mask, bool)[..., None], 0.0, value)
    value = value.reshape(B, S, M, C // M)
    off = lin("sampling_offsets", q).reshape(B, Lq, M, L, P, 2)
    logit = lin("attention_weights", q).reshape(B, Lq, M, L * P)
    logit = logit - logit.max(-1, keepdims=True)
    aw = np.exp(logit)
    aw = (aw / aw.sum(-1, keepdims=True)).reshape(B, Lq, M, L, P)
    shp = np.asarray(shapes, dtype)
    if ref.shape[-1] == 2:
        normalizer = np.stack([shp[:, 1], shp[:, 0]], -1)
        loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    elif ref.shape[-1] == 4:
        if use_4d_normalizer:
            normalizer = np.stack([shp[:, 1], shp[:, 0]], -1)
            loc = ref[:, :, None, :, None, :2] + off / normalizer[None, None, None, :, None, :] * \
                ref[:, :, None, :, None, 2:] * 0.5
        else:
            loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    else:
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(ref.shape[-1]))
    core = forward(np.ascontiguousarray(value), shapes, lsi, np.ascontiguousarray(loc), np.ascontiguousarray(aw))
    return lin("output_proj", np.asarray(core, dtype)), loc, aw

