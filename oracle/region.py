"""ORACLE -- test infrastructure only.  CPU restatement of the region encoder's point sampling
(VisionLLMv2/visionllmv2/model/region_encoder.py:24-47: point_sample = grid_sample(input, 2 * coords - 1); :135-140: the
masked mean of the sampled features).  torch's grid_sample IS the reference's arithmetic here; a plain numpy restatement
of the same bilinear / zeros / align_corners=False rule sits next to it so that the rule itself is pinned."""
import numpy as np


def point_sample(inp, coords):
    import torch
    import torch.nn.functional as F
    x, c = torch.as_tensor(inp), torch.as_tensor(coords)
    return F.grid_sample(x, 2.0 * c.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)


def masked_mean(sampled, valid):
    import torch
    v = torch.as_tensor(valid)
    feats = sampled.permute(0, 2, 1) * v.unsqueeze(-1)
    return (feats.sum(1) / v.sum(1).unsqueeze(-1)).nan_to_num()


def point_sample_numpy(inp, coords):
    """ix = ((2c - 1 + 1) * W - 1) / 2, corners floor / floor+1, a corner outside the map contributes nothing."""
    inp, coords = np.asarray(inp, np.float64), np.asarray(coords, np.float64)
    N, C, H, W = inp.shape
    P = coords.shape[1]
    out = np.zeros((N, C, P))
    for n in range(N):
        for p in range(P):
            ix = ((2 * coords[n, p, 0] - 1 + 1) * W - 1) / 2
            iy = ((2 * coords[n, p, 1] - 1 + 1) * H - 1) / 2
            x0, y0 = int(np.floor(ix)), int(np.floor(iy))
            for dy in (0, 1):
                for dx in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    if 0 <= xx < W and 0 <= yy < H:
                        wgt = (1 - abs(ix - xx)) * (1 - abs(iy - yy))
                        out[n, :, p] += wgt * inp[n, :, yy, xx]
    return out
