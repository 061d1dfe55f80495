"""Shared host logic of the two vision-encoder mirrors (InternViT, CLIP): parameter packing into the C ABI
descriptor, hidden-state allocation, the forward call."""
import ctypes

import torch

from . import _lib


def _require_bf16_cuda(name, t):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor: the native vision encoder has no CPU path")
    if t.dtype != torch.bfloat16:
        raise RuntimeError(f"{name} must be bfloat16 (got {t.dtype}); load the model with torch_dtype=torch.bfloat16 "
                           "as the reference does (train/train.py:375)")


class EncoderPlan:
    """Device pointers of every parameter, packed once and refreshed when a parameter changes."""

    def __init__(self):
        self.key = None
        self.desc = None
        self.layers = None
        self.keep = []  # tensors owned by the plan (padded / fused copies)

    @staticmethod
    def signature(params):
        return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)


def padded_patch_weight(conv_weight, kpad):
    """Conv2d weight [C,3,ps,ps] -> [C, kpad] (k = c*ps*ps + ky*ps + kx, zero padded)."""
    C = conv_weight.shape[0]
    w = conv_weight.detach().reshape(C, -1)
    out = torch.zeros((C, kpad), dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


def fold_norm_into_linear(w, b, gamma, beta, layernorm):
    """Operands of a linear layer with the norm in front of it folded in (vllm_gemm_bf16_ln, consumer side):
        linear(norm(x)) = r (x W'^T) - r mean colsum + bias'      W' = W diag(gamma)  (bf16),
        colsum_n = sum_k W'[n, k]  (fp32, from the ROUNDED W': the mean shift then cancels exactly for the weights in use),
        bias'_n = b_n + sum_k beta_k W[n, k]  (fp32).
    RMSNorm (layernorm=False) has no mean and no beta: colsum is None, bias' is b.  -> (w_ln, colsum, bias_ln)"""
    w32 = w.detach().float()
    w_ln = (w32 * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    colsum = w_ln.float().sum(1).contiguous() if layernorm else None
    bias_ln = b.detach().float().clone() if b is not None else None
    if beta is not None:
        shift = w32 @ beta.detach().float()
        bias_ln = shift if bias_ln is None else bias_ln + shift
    return w_ln, colsum, (bias_ln.contiguous() if bias_ln is not None else None)


def norm_folding_applies(hidden_size, intermediate_size, rms=False):
    """The encoder folds its norms into the GEMMs around them
      * at hidden size 1024 (ViT-L, InternViT-300M: rows of exactly four 256-column tiles, LayerNorm or RMSNorm), and
      * round 5: for RMSNorm at any hidden size of up to 16 column tiles that is a multiple of 8 (InternViT-6B: 3200 = 12.5 tiles;
        `rms=True`), where the C side keeps the launched norms for batches too small for the persistent GEMM schedule.
    The C side also asks for >= 1024 tokens per call; VLLM_LN_FOLD=0 keeps the norm launches (A/B).  The prepared operands are one
    more bf16 copy of the qkv and fc1 weights (InternViT-6B: 6.9 GB over 48 layers)."""
    import os
    if os.environ.get("VLLM_LN_FOLD", "1") == "0" or intermediate_size < 1024:
        return False
    return hidden_size == 1024 or (rms and hidden_size % 8 == 0 and hidden_size <= 4096 and os.environ.get("VLLM_LN_FOLD_WIDE", "1") != "0")


def kpad_for(patch):
    k = 3 * patch * patch
    return (k + 63) // 64 * 64


def run_encoder(desc, pixel_values, num_layers, hidden_size, keep=None):
    """Allocate hidden states and call vllm_vit_forward.  ``keep``: None = materialise all L+1 states (what the
    reference does with output_hidden_states=True), or an iterable of indices (negative allowed) to keep."""
    lib = _lib.lib()
    if pixel_values.dim() != 4 or pixel_values.shape[1] != 3:
        raise ValueError(f"wrong pixel_values size: {tuple(pixel_values.shape)}")
    if not pixel_values.is_cuda:
        raise RuntimeError("pixel_values must be a CUDA tensor: the native vision encoder has no CPU path")
    if pixel_values.shape[2] != desc.image or pixel_values.shape[3] != desc.image:
        raise ValueError(f"pixel_values must be {desc.image}x{desc.image}, got {tuple(pixel_values.shape[2:])}")
    if pixel_values.dtype not in (torch.bfloat16, torch.float32):
        pixel_values = pixel_values.to(torch.bfloat16)
    pixel_values = pixel_values.contiguous()
    desc.pixel_is_f32 = 1 if pixel_values.dtype == torch.float32 else 0
    n = pixel_values.shape[0]
    g = desc.image // desc.patch
    S = g * g + 1
    dev = pixel_values.device
    L = num_layers
    if keep is None:
        want = set(range(L + 1))
    else:
        want = {(i + L + 1) % (L + 1) for i in keep} | {L}
    states = [torch.empty((n, S, hidden_size), dtype=torch.bfloat16, device=dev) if i in want else None
              for i in range(L + 1)]
    # Tiles never interact inside the encoder, so a batch CAN be run as independent chunks on separate streams (opt-in:
    # set_encoder_chunks / VLLM_ENCODER_CHUNKS), the idea being that kernels of the other chunk fill the CUs a GEMM / attention
    # launch leaves idle in its last round of tiles.  Same kernels, same per-tile arithmetic: results are bit-identical
    # (tested).  Measured on the bench workload (40 tiles ViT-L): 24.08 ms per step with two half-batches against 22.65 ms
    # with one sequence -- the half-size GEMMs quantise worse than the overlap recovers -- so the default is ONE chunk.
    chunks = encoder_chunks(n)
    bounds = [n * c // chunks for c in range(chunks + 1)]
    row_bytes = S * hidden_size * 2
    px_bytes = 3 * desc.image * desc.image * pixel_values.element_size()
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        for c in range(chunks):
            lo, nc = bounds[c], bounds[c + 1] - bounds[c]
            if nc == 0:
                continue
            ptrs = (ctypes.c_void_p * (L + 1))(*[s.data_ptr() + lo * row_bytes if s is not None else None for s in states])
            ws_bytes = lib.vllm_vit_workspace_bytes(ctypes.byref(desc), nc)
            if ws_bytes < 0:
                raise RuntimeError("vllm_vit_workspace_bytes: " + lib.vllm_last_error().decode())
            ws = _lib.workspace(dev, ws_bytes, slot=c)
            stream = main if c == 0 else _side_stream(dev, c)
            if c > 0:
                stream.wait_stream(main)   # inputs (and the workspace's previous users) are ordered on the caller's stream
            with torch.cuda.stream(stream):
                _lib.check(lib.vllm_vit_forward(ctypes.byref(desc), ctypes.c_void_p(pixel_values.data_ptr() + lo * px_bytes), nc,
                                                ptrs, _lib.ptr(ws), ws_bytes, _lib.current_stream(dev)), "vllm_vit_forward")
        for c in range(1, chunks):
            main.wait_stream(_side_stream(dev, c))
    return states


_ENCODER_CHUNKS = {"value": None}
_SIDE_STREAMS = {}


def set_encoder_chunks(k):
    """1 (default, also None): one launch sequence per batch; k: k chunks of tiles on k streams (measured slower)."""
    old = _ENCODER_CHUNKS["value"]
    _ENCODER_CHUNKS["value"] = k
    return old


def encoder_chunks(n):
    import os
    k = _ENCODER_CHUNKS["value"]
    if k is None:
        env = os.environ.get("VLLM_ENCODER_CHUNKS")
        k = int(env) if env else 1
    return max(1, min(int(k), n, 4))


def _side_stream(dev, c):
    key = (str(dev), c)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class LazyHiddenStates(tuple):
    """Tuple of L+1 hidden states; entries that were not materialised are None (only with keep=...)."""


def model_output(last_hidden_state, pooler_output, hidden_states, return_dict=True):
    if not return_dict:
        return (last_hidden_state, pooler_output) + ((hidden_states,) if hidden_states is not None else ())
    try:
        from transformers.modeling_outputs import BaseModelOutputWithPooling
        return BaseModelOutputWithPooling(last_hidden_state=last_hidden_state, pooler_output=pooler_output,
                                          hidden_states=hidden_states, attentions=None)
    except Exception:  # transformers not importable: a minimal stand-in with the attributes the caller reads
        from types import SimpleNamespace
        return SimpleNamespace(last_hidden_state=last_hidden_state, pooler_output=pooler_output,
                               hidden_states=hidden_states, attentions=None)
