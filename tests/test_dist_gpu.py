"""The visual-token all-gather on the RCCL (``nccl``) backend, one process per GPU (skipped on boxes with fewer than 2 GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from visionllm_amd.dist import all_gather_visual_tokens
    res = {}
    for name, n in (("equal", 2), ("ragged", 3 if rank == 0 else 1)):
        tok = (torch.arange(n * 4 * 64, dtype=torch.float32, device=dev).reshape(n, 4, 64) + 1000 * rank).to(torch.bfloat16)
        out, counts = all_gather_visual_tokens(tok)
        out2, _ = all_gather_visual_tokens(tok, counts=counts, async_op=True).wait()
        out3, _ = all_gather_visual_tokens(tok, algo="direct")
        out4, _ = all_gather_visual_tokens(tok, counts=counts, async_op=True, algo="direct").wait()
        torch.cuda.synchronize()
        assert torch.equal(out, out2) and torch.equal(out, out3) and torch.equal(out, out4)
        res[name] = (out.float().cpu(), counts)
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_all_gather_visual_tokens_nccl_equal_and_ragged():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    for name, cnt in (("equal", [2, 2]), ("ragged", [3, 1])):
        a, b = res[0][1][name], res[1][1][name]
        assert a[1] == cnt and b[1] == cnt and torch.equal(a[0], b[0]) and a[0].shape[0] == sum(cnt)
        assert a[0][cnt[0], 0, 0] == 1000.0   # rank 1's first tile follows rank 0's tiles
