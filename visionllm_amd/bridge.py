"""Visual-token projector ("vl_bridge") + hidden-state select + pixel-shuffle: boundary B2.

Mirrors VisionLLMv2/visionllmv2/model/modeling_visionllmv2.py:157-190 (construction: ``linear``,
``internvl_mlp``/``internvl`` = LayerNorm, Linear, GELU, Linear; ``mlp{N}x_gelu``), :381-392 (``pixel_shuffle``)
and :569-579 (select ``hidden_states[vis_output_layer][:, 1:]``, optional pixel-shuffle, bridge).

``build_vl_bridge`` returns an ``nn.Linear`` / ``nn.Sequential`` subclass with the SAME child indices, so
``vl_bridge.{idx}.weight/bias`` state-dict keys (and ``vl_bridge.bin`` files, :185-190) load unchanged; its
``forward(x)`` takes what the reference passes (``image_features_ori`` [n, T, C_in]).  ``project_hidden_state`` is the
fused path used by our own pipeline: it reads the selected hidden state [n, 1+T, C] directly (CLS rows are skipped
inside the GEMM loader / pixel-shuffle gather) and writes [n, T', C_llm].
"""
import ctypes
import re

import torch
from torch import nn

from . import _lib


def pixel_shuffle(x, scale_factor=0.5):
    """[n, w, h, c] -> [n, w/2, h/2, 4c] (modeling_visionllmv2.py:381-392), one gather kernel."""
    if scale_factor != 0.5:
        raise NotImplementedError("only scale_factor=0.5 is used by VisionLLMv2")
    if not x.is_cuda or x.dtype != torch.bfloat16:
        raise RuntimeError("pixel_shuffle: bf16 CUDA tensor required (no CPU path)")
    n, w, h, c = x.shape
    if w != h:
        raise ValueError("pixel_shuffle: square token grid expected")
    x = x.contiguous()
    out = torch.empty((n, w // 2, h // 2, 4 * c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vllm_pixel_shuffle_bf16(_lib.ptr(x), w * h * c, c, 0, _lib.ptr(out), n, w, c,
                                                      _lib.current_stream(x.device)), "vllm_pixel_shuffle_bf16")
    return out


def _desc_for(mod, kind, pixel_shuffle_flag, skip_cls):
    P = _lib.ptr
    if kind == "linear":
        lin = [mod]
        ln = None
        k = _lib.BRIDGE_LINEAR
    elif kind in ("internvl_mlp", "internvl"):
        ln, lin, k = mod[0], [mod[1], mod[3]], _lib.BRIDGE_INTERNVL_MLP
    else:
        lin, ln, k = [m for m in mod if isinstance(m, nn.Linear)], None, _lib.BRIDGE_MLP_GELU
    for m in lin + ([ln] if ln is not None else []):
        for p in m.parameters():
            if not p.is_cuda or p.dtype != torch.bfloat16:
                raise RuntimeError("vl_bridge parameters must be bf16 CUDA tensors (no CPU path)")
    d = _lib.VllmBridgeDesc(kind=k, depth=len(lin), in_features=lin[0].in_features, out_features=lin[-1].out_features,
                            pixel_shuffle=int(pixel_shuffle_flag), skip_cls=int(skip_cls),
                            ln_eps=ln.eps if ln is not None else 0.0, ln_w=P(ln.weight) if ln is not None else None,
                            ln_b=P(ln.bias) if ln is not None else None)
    for i, m in enumerate(lin):
        d.w[i] = m.weight.data_ptr()
        d.b[i] = m.bias.data_ptr() if m.bias is not None else None
    return d


def _run(desc, hidden, n, T, C):
    lib = _lib.lib()
    _lib.check_struct_layouts()
    T_out = T // 4 if desc.pixel_shuffle else T
    out = torch.empty((n, T_out, desc.out_features), dtype=torch.bfloat16, device=hidden.device)
    with torch.cuda.device(hidden.device):
        wsb = lib.vllm_bridge_workspace_bytes(ctypes.byref(desc), n, T)
        ws = _lib.workspace(hidden.device, max(wsb, 1))
        _lib.check(lib.vllm_bridge_forward(ctypes.byref(desc), _lib.ptr(hidden), n, T, C, _lib.ptr(out), _lib.ptr(ws),
                                           wsb, _lib.current_stream(hidden.device)), "vllm_bridge_forward")
    return out


class _BridgeMixin:
    vl_bridge_type = "linear"

    def _native(self, x, pixel_shuffle_flag=False, skip_cls=False):
        if not x.is_cuda or x.dtype != torch.bfloat16:
            raise RuntimeError("vl_bridge: bf16 CUDA input required (no CPU path)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # the reference trains this slot (tune_vl_bridge); the native kernels are forward-only and their output has no
            # grad_fn: refuse loudly instead of silently cutting the gradient
            raise RuntimeError("vl_bridge (native): forward-only kernels -- call under torch.no_grad() or freeze the "
                               "projector and its input (requires_grad_(False)); training the projector needs the torch modules")
        x = x.contiguous()
        n, rows, C = x.shape
        T = rows - 1 if skip_cls else rows
        return _run(_desc_for(self, self.vl_bridge_type, pixel_shuffle_flag, skip_cls), x, n, T, C)

    def forward(self, x):  # what modeling_visionllmv2.py:579 calls
        lead = x.shape[:-2] if x.dim() > 3 else None
        if x.dim() == 2:
            return self._native(x[None])[0]
        if lead is not None:
            x = x.reshape(-1, *x.shape[-2:])
        y = self._native(x)
        return y.reshape(*lead, *y.shape[-2:]) if lead is not None else y

    def project_hidden_state(self, hidden_state, use_pixelshuffle=False):
        """hidden_state [n, 1+T, C] (a hidden_states[vis_output_layer] entry) -> [n, T or T/4, C_llm]."""
        return self._native(hidden_state, use_pixelshuffle, True)


class NativeBridgeLinear(_BridgeMixin, nn.Linear):
    pass


class NativeBridgeSequential(_BridgeMixin, nn.Sequential):
    pass


def build_vl_bridge(vl_bridge_type, v_hidden_size, l_hidden_size, use_pixelshuffle=False):
    """Same construction logic as modeling_visionllmv2.py:160-184."""
    v = v_hidden_size * 4 if use_pixelshuffle else v_hidden_size
    if vl_bridge_type == "linear":
        m = NativeBridgeLinear(v, l_hidden_size)
    elif vl_bridge_type in ("internvl_mlp", "internvl"):
        m = NativeBridgeSequential(nn.LayerNorm(v), nn.Linear(v, l_hidden_size), nn.GELU(),
                                   nn.Linear(l_hidden_size, l_hidden_size))
    else:
        mm = re.match(r"^mlp(\d+)x_gelu*", vl_bridge_type)
        if not mm:
            raise NotImplementedError(f"{vl_bridge_type} not supported yet.")
        depth = int(mm.group(1))
        if depth > 4:
            raise NotImplementedError("mlp depth > 4")
        mods = [nn.Linear(v, l_hidden_size)]
        for _ in range(1, depth):
            mods += [nn.GELU(), nn.Linear(l_hidden_size, l_hidden_size)]
        m = NativeBridgeSequential(*mods)
    m.vl_bridge_type = vl_bridge_type
    return m


def select_and_project(hidden_states, vl_bridge, vis_output_layer=-2, use_pixelshuffle=False):
    """modeling_visionllmv2.py:569-579 in one call: hs[layer][:,1:] (-> pixel_shuffle) -> vl_bridge."""
    return vl_bridge.project_hidden_state(hidden_states[vis_output_layer], use_pixelshuffle)
