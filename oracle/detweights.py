"""ORACLE -- test infrastructure only (imported by oracle/gen_golden_fullsize.py and by tests/, never by the product).

Deterministic, device-independent pseudo-random tensors for the FULL-SIZE parity fixtures (BASELINE.json configs[1] / [2]).

The full-size models have 0.3 / 5.9 G parameters: a fixture cannot carry them, and ``torch.manual_seed`` streams differ between
the CPU (where the reference runs when the fixture is made) and the GPU (where the native path is tested).  So every parameter
is a pure function of (tensor name, flat index): an integer hash evaluated with int64 tensor arithmetic -- exact on any device --
mapped to a uniform value in [-amp, amp) with 16 bits of resolution and rounded to bf16.  The fixture generator and the GPU test
call the SAME function and get bit-identical bf16 tensors (checked by a CRC in the fixture).

Distribution: uniform with the standard deviation the reference's initialisers use (0.02 for the matrices:
``amp = 0.02 * sqrt(3)``); norms / LayerScale / biases are perturbed around their initial values so that a swapped or missing
parameter shows up.
"""
import zlib

import torch

_MASK = 0xFFFFFFFF


def _name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) & _MASK


def hash_uniform(name: str, shape, amp: float, center: float = 0.0, device="cpu", dtype=torch.bfloat16, chunk: int = 1 << 24):
    """Tensor of `shape` with value(i) = center + amp * u(i), u in [-1, 1) a 16-bit hash of (name, i); rounded to `dtype`."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=dtype, device=device)
    seed = _name_seed(name)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        h = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = (h * 2654435761 + seed) & _MASK
        h = h ^ (h >> 15)
        h = (h * 2246822519) & _MASK
        h = h ^ (h >> 13)
        h = (h * 3266489917) & _MASK
        h = h ^ (h >> 16)
        u = ((h & 0xFFFF) - 32768).to(torch.float32) * (1.0 / 32768.0)   # exact
        out[lo:hi] = (u * amp + center).to(dtype)                           # fp32 multiply-add, then ONE rounding (RNE everywhere)
    return out.reshape(*shape)


MAT_AMP = 0.02 * 3 ** 0.5


def intern_vit_param(name: str, shape, device="cpu", dtype=torch.bfloat16):
    """Parameter `name` (state-dict key of InternVisionModel) of the deterministic InternViT weight set."""
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or "q_norm" in name or "k_norm" in name:
        return hash_uniform(name, shape, 0.2, 1.0, device, dtype)
    if name.endswith("ls1") or name.endswith("ls2"):
        return hash_uniform(name, shape, 0.05, 0.1, device, dtype)
    if name.endswith(".bias"):
        return hash_uniform(name, shape, 0.05, 0.0, device, dtype)
    if "class_embedding" in name or "position_embedding" in name:
        return hash_uniform(name, shape, 0.5, 0.0, device, dtype)
    return hash_uniform(name, shape, MAT_AMP, 0.0, device, dtype)


def clip_param(name: str, shape, device="cpu", dtype=torch.bfloat16):
    """Parameter `name` (state-dict key of transformers.CLIPVisionModel) of the deterministic CLIP ViT weight set."""
    if "layer_norm" in name or "layrnorm" in name:
        return hash_uniform(name, shape, 0.2, 1.0 if name.endswith("weight") else 0.0, device, dtype)
    if name.endswith(".bias"):
        return hash_uniform(name, shape, 0.05, 0.0, device, dtype)
    if "embedding" in name and "patch" not in name:
        return hash_uniform(name, shape, 0.5, 0.0, device, dtype)
    return hash_uniform(name, shape, MAT_AMP, 0.0, device, dtype)


def bridge_param(name: str, shape, device="cpu", dtype=torch.bfloat16):
    """Parameter of the vl_bridge (keys ``vl_bridge.<idx>.weight`` / ``.bias``); LayerNorm of internvl_mlp = index 0, 1-d weight."""
    if name.endswith("weight") and len(shape) == 1:
        return hash_uniform(name, shape, 0.2, 1.0, device, dtype)
    if name.endswith("bias"):
        return hash_uniform(name, shape, 0.05, 0.0, device, dtype)
    return hash_uniform(name, shape, MAT_AMP, 0.0, device, dtype)


def pixels(name: str, n, size, device="cpu", dtype=torch.bfloat16):
    """n normalised image tiles [n, 3, size, size] (CLIP-normalised pixels span about +-2)."""
    return hash_uniform(name, (n, 3, size, size), 2.0, 0.0, device, dtype)


def canonical(name: str) -> str:
    """State-dict key without the wrapper prefix that differs between transformers versions (4.x: ``vision_model.``)."""
    return name[len("vision_model."):] if name.startswith("vision_model.") else name


def fill_module_(module, fn, device=None):
    """Overwrite every parameter of `module` (in place, keeping its dtype) with fn(name, shape); returns a CRC of the bf16 bits."""
    crc = 0
    with torch.no_grad():
        for name, p in sorted(module.named_parameters(), key=lambda kv: canonical(kv[0])):
            v = fn(canonical(name), tuple(p.shape), device=p.device if device is None else device)
            p.copy_(v.to(p.dtype))
            crc = zlib.crc32(v.reshape(-1)[:4096].view(torch.int16).cpu().numpy().tobytes(), crc)
    return crc & _MASK
