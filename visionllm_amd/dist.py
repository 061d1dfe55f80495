"""Data-parallel sharding of images and the all-gather of visual tokens before the LLM stage.

The reference is pure data parallel (torchrun + DistributedSampler, VisionLLMv2/visionllmv2/train/train.py:586-588);
every rank feeds its own LLM replica, so there is no token exchange in the reference.  The north star adds ONE
exchange step: an all-gather of the projector output ``[n_tiles_r, T, C_llm]`` (bf16, 84-189 MB per rank at 8 images
x 5 tiles) over RCCL/xGMI.  Tiles per rank vary (1-7 tiles per image), so the counts are exchanged first and the
payload is padded to the maximum (``all_gather_into_tensor`` needs equal shards).  On ROCm the ``nccl`` backend IS
RCCL; ``gloo`` is used by the CPU tests.
"""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_images(tiles_per_image: Sequence[int], world_size: int) -> List[List[int]]:
    """Assign whole images to ranks, balancing TILE counts (greedy longest-processing-time), keeping every
    image's tiles on one rank so ``split_sizes`` stay contiguous (modeling_visionllmv2.py:563, 587-588).
    Returns image indices per rank (each list sorted)."""
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    order = sorted(range(len(tiles_per_image)), key=lambda i: (-tiles_per_image[i], i))
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += tiles_per_image[i]
    return [sorted(x) for x in out]


class GatherHandle:
    """Result of an asynchronous ``all_gather_visual_tokens``: call ``wait()`` to get (tokens, tiles per rank)."""

    def __init__(self, work, out, counts, mx):
        self._work, self._out, self._counts, self._mx = work, out, counts, mx

    def wait(self):
        if self._work is not None:
            self._work.wait()   # makes the current stream wait for the collective (no host sync with NCCL/RCCL)
            self._work = None
        out, counts, mx = self._out, self._counts, self._mx
        if len(counts) == 1 or all(c == mx for c in counts):
            return out, counts
        return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(len(counts))], 0), counts


class _Works:
    """A set of point-to-point requests behaving like one collective work handle."""

    def __init__(self, reqs):
        self._reqs = reqs

    def wait(self):
        for r in self._reqs:
            r.wait()


def _direct_all_gather(out, shard, mx, group, async_op):
    """All peers at once: every rank posts one send per peer and one receive per peer (batched point-to-point), the
    pattern SURVEY.md section 5 argues for on the xGMI full mesh (7 links x ~153 GB/s per GPU, all busy at the same time,
    84-189 MB per message) instead of a ring that is bound by one link.  RCCL runs the batch on its own stream."""
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    out[rank * mx:(rank + 1) * mx].copy_(shard)
    ops = []
    for step in range(1, ws):                       # staggered peers: rank r talks to r+step / r-step in round `step`
        dst, src = (rank + step) % ws, (rank - step) % ws
        ops.append(dist.P2POp(dist.isend, shard, dist.get_global_rank(group, dst) if group is not None else dst, group))
        ops.append(dist.P2POp(dist.irecv, out[src * mx:(src + 1) * mx], dist.get_global_rank(group, src) if group is not None else src, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return _Works(reqs)
    for r in reqs:
        r.wait()
    return None


def all_gather_visual_tokens(tokens: torch.Tensor, group=None, counts: Sequence[int] = None, async_op: bool = False,
                             algo: str = None):
    """tokens [n_tiles_r, T, C] -> ([sum_r n_tiles_r, T, C] in rank order, tiles per rank).

    ``algo``: "collective" (default; RCCL's all_gather_into_tensor picks its own algorithm) or "direct" (batched
    point-to-point to every peer at once, see _direct_all_gather); the environment variable VLLM_ALLGATHER overrides the
    default.

    One small all-gather of the counts (skipped when the caller passes ``counts`` -- e.g. it sharded the images itself
    with ``shard_images`` -- which also avoids the host sync of reading them back), one large all-gather of the
    (padded) payload.  ``async_op=True`` returns a ``GatherHandle`` immediately: the collective runs on RCCL's stream
    over xGMI while the caller keeps enqueuing independent work (the det-head kernels) on the compute stream."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        res = (tokens, [tokens.shape[0]])
        return GatherHandle(None, res[0], res[1], tokens.shape[0]) if async_op else res
    ws = dist.get_world_size(group)
    if counts is None:
        n = torch.tensor([tokens.shape[0]], dtype=torch.int64, device=tokens.device)
        cnt = torch.empty(ws, dtype=torch.int64, device=tokens.device)
        dist.all_gather_into_tensor(cnt, n, group=group)
        counts_l = [int(c) for c in cnt.tolist()]
    else:
        counts_l = [int(c) for c in counts]
        if len(counts_l) != ws or counts_l[dist.get_rank(group)] != tokens.shape[0]:
            raise ValueError("all_gather_visual_tokens: `counts` must list the tile count of every rank")
    mx = max(counts_l)
    T, C = tokens.shape[1], tokens.shape[2]
    if tokens.shape[0] != mx:
        pad = torch.zeros((mx, T, C), dtype=tokens.dtype, device=tokens.device)
        pad[: tokens.shape[0]] = tokens
        tokens = pad
    out = torch.empty((ws * mx, T, C), dtype=tokens.dtype, device=tokens.device)
    algo = algo or os.environ.get("VLLM_ALLGATHER", "collective")
    if algo not in ("collective", "direct"):
        raise ValueError(f"all_gather_visual_tokens: unknown algo {algo!r}")
    if algo == "direct":
        work = _direct_all_gather(out, tokens.contiguous(), mx, group, async_op)
    else:
        work = dist.all_gather_into_tensor(out, tokens.contiguous(), group=group, async_op=async_op)
    h = GatherHandle(work if async_op else None, out, counts_l, mx)
    return h if async_op else h.wait()
