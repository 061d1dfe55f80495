"""Seeded synthetic MSDA inputs shared by tests and bench (numpy; no reference access at run time).

``encoder_like``: sampling locations = encoder reference grid (pixel centres of every level, as
modeling_ov_grounding_dino_mask_dn.py:1579-1606 builds them) + offsets following the init pattern of the
sampling_offsets bias (:683-697: unit direction per head scaled by (k+1), in pixels of the target level) plus
Gaussian jitter.  ``stress``: uniform locations in [-0.05, 1.05] (exercises rejection and border corners).
"""
import math

import numpy as np

CFG4_SHAPES = [(168, 168), (84, 84), (42, 42), (21, 21)]  # 1344x1344 input, strides 8/16/32/64


def level_start_index(shapes):
    a = np.array([h * w for h, w in shapes], dtype=np.int64)
    return np.concatenate([[0], np.cumsum(a)[:-1]]).astype(np.int64)


def reference_grid(shapes):
    """[S, 2] (x, y) normalised pixel centres, levels concatenated (the encoder's own queries)."""
    pts = []
    for h, w in shapes:
        ys, xs = np.meshgrid((np.arange(h, dtype=np.float32) + 0.5) / h, (np.arange(w, dtype=np.float32) + 0.5) / w,
                             indexing="ij")
        pts.append(np.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    return np.concatenate(pts, 0).astype(np.float32)


def make_inputs(B, M, D, shapes, P, Lq=None, mode="encoder_like", seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    L = len(shapes)
    S = int(sum(h * w for h, w in shapes))
    value = rng.standard_normal((B, S, M, D), dtype=np.float32)
    if mode in ("encoder_like", "encoder_wide"):   # encoder_wide: the same pattern with 3x the offsets and jitter (trained heads)
        spread = 3.0 if mode == "encoder_wide" else 1.0
        ref = reference_grid(shapes)  # [S,2]
        if Lq is None or Lq == S:
            Lq = S
            q_ref = ref
        else:
            q_ref = rng.random((Lq, 2), dtype=np.float32)
        th = np.arange(M, dtype=np.float32) * (2.0 * math.pi / M)
        g = np.stack([np.cos(th), np.sin(th)], -1)
        g = g / np.abs(g).max(-1, keepdims=True)  # [M,2]
        off = g[:, None, None, :] * (np.arange(P, dtype=np.float32) + 1)[None, None, :, None]  # [M,1,P,2] pixels
        off = np.broadcast_to(off, (M, L, P, 2)).copy()
        wh = np.array([[w, h] for h, w in shapes], dtype=np.float32)  # [L,2]
        loc = np.empty((B, Lq, M, L, P, 2), dtype=np.float32)
        for b in range(B):
            jit = rng.standard_normal((Lq, M, L, P, 2), dtype=np.float32) * 0.5
            loc[b] = q_ref[:, None, None, None, :] + spread * (off[None] + jit) / wh[None, None, :, None, :]
    elif mode == "stress":
        Lq = Lq or 64
        loc = rng.random((B, Lq, M, L, P, 2), dtype=np.float32) * 1.1 - 0.05
        flat = loc.reshape(-1)
        flat[0::97] = 0.0
        flat[1::101] = 1.0
        flat[2::103] = 0.5
    else:
        raise ValueError(mode)
    logits = rng.standard_normal((B, Lq, M, L * P), dtype=np.float32)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    attw = (e / e.sum(-1, keepdims=True)).reshape(B, Lq, M, L, P).astype(np.float32)
    shapes_a = np.array(shapes, dtype=np.int64)
    return dict(value=value.astype(dtype), shapes=shapes_a, lsi=level_start_index(shapes), loc=loc.astype(dtype),
                attw=attw.astype(dtype))
