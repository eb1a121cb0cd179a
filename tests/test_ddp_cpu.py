"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: sample sharding and the bucketed
gradient all-reduce of tactilesr_amd.ddp (the N>1 path of bench.py --mode train)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tactilesr_amd import ddp
    r, w, _ = ddp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                  # replicas start different on purpose
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    sync = ddp.GradSync(model.parameters(), n_buckets=3)
    sync.broadcast_parameters(0)
    w0 = torch.cat([p.detach().flatten() for p in model.parameters()])
    # sharded batch: rank-local loss on its contiguous slice
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 3, 8, 8, generator=g)
    a, b = ddp.shard_batch(6, rank, world)
    loss = model(x[a:b]).pow(2).mean()
    loss.backward()
    local = [p.grad.clone() for p in model.parameters()]
    sync()
    synced = torch.cat([p.grad.flatten() for p in model.parameters()])
    gathered = [torch.zeros_like(synced) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([t.flatten() for t in local]))
    expect = sum(gathered) / world
    ok = torch.allclose(synced, expect, atol=1e-7) and (a, b) == (3 * rank, 3 * rank + 3)
    covered = sorted(sync.buckets) == sync.buckets and sync.buckets[0][0] == 0 and \
        sync.buckets[-1][1] == synced.numel() and all(sync.buckets[i][1] == sync.buckets[i + 1][0]
                                                      for i in range(len(sync.buckets) - 1))
    q.put((rank, bool(ok), bool(covered), w0.sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res
    assert abs(res[0][3] - res[1][3]) < 1e-6          # broadcast made the replicas identical


def test_shard_batch_covers_everything():
    from tactilesr_amd import ddp
    for n in (1, 7, 8, 65536):
        for world in (1, 2, 8):
            spans = [ddp.shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
