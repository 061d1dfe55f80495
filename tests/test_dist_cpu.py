"""world_size-2 gloo tests of the data-parallel path (CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visionllm_amd.dist import all_gather_visual_tokens, shard_images


def test_shard_images_balances_tiles_and_keeps_images_whole():
    tiles = [5, 1, 7, 3, 5, 5, 2, 4]
    sh = shard_images(tiles, 4)
    assert sorted(i for r in sh for i in r) == list(range(8))
    loads = [sum(tiles[i] for i in r) for r in sh]
    assert max(loads) - min(loads) <= 2
    assert shard_images([5] * 8, 8) == [[i] for i in range(8)]
    assert shard_images([], 2) == [[], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ragged, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = (3 if rank == 0 else 1) if ragged else 2
    torch.manual_seed(rank)
    tok = (torch.arange(n * 4 * 8, dtype=torch.float32).reshape(n, 4, 8) + 1000 * rank).to(torch.bfloat16)
    out, counts = all_gather_visual_tokens(tok)
    # the asynchronous form with caller-provided counts (no count exchange, no host sync) must give the same result
    h = all_gather_visual_tokens(tok, counts=counts, async_op=True)
    out2, counts2 = h.wait()
    assert counts2 == counts and torch.equal(out2, out)
    # the direct (all peers at once, batched point-to-point) algorithm gives the same tensor, blocking and asynchronous
    out3, counts3 = all_gather_visual_tokens(tok, algo="direct")
    out4, _ = all_gather_visual_tokens(tok, counts=counts, async_op=True, algo="direct").wait()
    assert counts3 == counts and torch.equal(out3, out) and torch.equal(out4, out)
    # the bench's pipeline across steps: a step's collective is waited for one step later, the last one is drained at the end;
    # several collectives may be in flight and every one must deliver ITS step's tokens
    for algo in ("collective", "direct"):
        pending, got = None, []
        for stepi in range(4):
            t = tok + stepi
            h = all_gather_visual_tokens(t, counts=counts, async_op=True, algo=algo)
            prev, pending = pending, h
            if prev is not None:
                got.append(prev.wait()[0])
        got.append(pending.wait()[0])
        for stepi, g in enumerate(got):
            assert torch.equal(g, (out.float() + stepi).to(torch.bfloat16)), (algo, stepi)
    q.put((rank, out.float(), counts))
    dist.destroy_process_group()


def _run(ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, ragged, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    return res


def test_all_gather_equal_shards():
    res = _run(False)
    for _, out, counts in res:
        assert counts == [2, 2] and out.shape == (4, 4, 8)
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][1][2, 0, 0] == 1000.0  # rank 1's first tile follows rank 0's two tiles


def test_all_gather_ragged_shards():
    res = _run(True)
    for _, out, counts in res:
        assert counts == [3, 1] and out.shape == (4, 4, 8)
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][1][3, 0, 0] == 1000.0
