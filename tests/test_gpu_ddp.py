"""Two data-parallel ranks of the REAL HIP engine on one MI355X (two processes sharing the card, gloo backend -- it
moves CUDA tensors through the host; RCCL cannot put two ranks on one device and the pool leases one GPU).  What it
pins: the gradient arena + in-backward bucket all-reduce of tactilesr_amd.ddp on the engine's own backward, across real
process boundaries: synced gradient = mean over ranks of the per-shard fp64 ORACLE gradients (each rank's shard, its
own BatchNorm batch statistics, the device's ReLU pattern: SURVEY.md section 5), max-norm 1e-5 on every parameter;
every bucket enqueued before backward returns from the second step on, identical weights on both ranks after
the fused Adam step, parameter AND buffer broadcast at start."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import tactilesr_amd
    from tactilesr_amd import ddp, optim
    ddp.init_distributed("gloo")
    torch.cuda.set_device(0)
    cfg = dict(patternFeatureExtraLayerCnt=1)
    torch.manual_seed(1000 + rank)                        # replicas start different on purpose
    m = tactilesr_amd.TactileSR(**cfg).cuda().train()
    with torch.no_grad():
        for bn in [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]:
            bn.running_mean.fill_(0.25 * (rank + 1))
    sync = ddp.GradSync(m)
    sync.broadcast_parameters(0)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    LR, HR = torch.rand(6, 3, 4, 4, generator=g) * 8, torch.rand(6, 1, 40, 40, generator=g) * 25
    a, b = ddp.shard_batch(6, rank, world)
    x, y = LR[a:b].cuda(), HR[a:b].cuda()
    # What the synced gradient must equal: the MEAN over ranks of the CPU oracle's gradient on each
    # rank's own shard -- fp64, train-mode BatchNorm on that shard's batch statistics (rank-local BN), evaluated on the
    # ReLU pattern this rank's device forward took (tests/_gradcheck.py).  A fresh replica with the same weights gives
    # the pattern; its pattern must itself agree with the fp64 oracle's up to rounding-zero flips.
    import _gradcheck as GC
    ref = tactilesr_amd.TactileSR(**cfg).cuda().train()
    ref.load_state_dict(sd0)
    eng = ref.train_engine()
    eng.keep_ctx = True
    ref_loss = F.mse_loss(ref(x), y)
    ref_loss.backward()
    masks = {k: v.cpu() for k, v in eng.activation_masks(eng.last_ctx).items()}
    sd_cpu = {k: v.detach().cpu() for k, v in sd0.items()}
    l64, _, _, pre64 = GC.oracle_grads(sd_cpu, LR[a:b], HR[a:b], record=True)
    flips = GC.check_pattern(masks, pre64)
    _, g64m, _, _ = GC.oracle_grads(sd_cpu, LR[a:b], HR[a:b], masks=masks)
    names = [k for k, _ in m.named_parameters()]
    assert set(names) == set(g64m)
    local = torch.cat([g64m[k].flatten() for k in names])                 # fp64, CPU: gloo moves it as is
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = {}
    off = 0
    mean = sum(gathered) / world
    for k in names:
        n = g64m[k].numel()
        expect[k] = mean[off:off + n].view_as(g64m[k])
        off += n
    loss_ok = abs(float(ref_loss) - l64) < 1e-5 * abs(l64)
    opt = optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-2)
    res = {"rank": rank, "bn_mean": float(next(iter(sd0[k] for k in sd0 if k.endswith("running_mean")))[0]),
           "w0": float(torch.cat([v.flatten().double() for k, v in sd0.items() if v.is_floating_point()]).sum())}
    ok, events = [], []
    for step in range(2):
        m.load_state_dict(sd0)                            # same weights and BN statistics every step
        opt.zero_grad()
        sync.events.clear()
        F.mse_loss(m(x), y).backward()
        events.append(list(sync.events))
        sync.finish()
        try:
            worst = GC.check_grads({k: p.grad for k, p in m.named_parameters()}, expect, tol=1e-5)
            ok.append(True)
        except AssertionError as e:
            worst = (float("nan"), str(e)[:300])
            ok.append(False)
    opt.step()
    torch.cuda.synchronize()
    w1 = torch.cat([p.detach().flatten() for p in m.parameters()])
    both = [torch.zeros_like(w1) for _ in range(world)]
    dist.all_gather(both, w1)
    res.update(ok=ok, events=events, worst=worst, flips=flips, loss_ok=loss_ok, nb=len(m.train_engine().arena.buckets),
               same_weights=bool(torch.equal(both[0], both[1])), moved=bool(not torch.equal(w1, torch.cat(
                   [sd0[k].flatten() for k, _ in m.named_parameters()]))))
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_engine_gradsync_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["w0"] == res[1]["w0"] and res[1]["bn_mean"] == 0.25            # params + buffers = rank 0's
    for r in res:
        assert r["ok"] == [True, True] and r["loss_ok"], r
        print(f"[ddp vs per-shard oracle] rank {r['rank']}: {r['flips']} ReLU flips vs fp64 on its shard; synced gradient "
              f"vs mean of the per-shard fp64 oracle gradients: worst max-norm error {r['worst'][0]:.2e} ({r['worst'][1]})")
        nb = r["nb"]
        assert r["events"][0] == []                                              # first step: arena laid out at its end
        assert r["events"][1] == [("enqueue", k) for k in range(nb)], r["events"] # then: all buckets from inside backward
        assert r["same_weights"] and r["moved"]


def test_bench_train_two_ranks_fresh_subprocess_gloo():
    """`python bench.py --gpus 2 --mode train --batch 64` as a FRESH process (the way the driver starts it, outside
    torchrun): bench.py spawns its two ranks itself, both run the real HIP engine on this box's one GPU, gradients go
    over gloo (TSR_BENCH_DIST_BACKEND=gloo; RCCL needs one GPU per rank).  Checks the contract line: n_gpus == 2, both
    ranks' losses finite, the data-parallel timing fields the first multi-GPU run will be read by (dist_world_size,
    comm_wait_ms, per-bucket enqueue offsets inside backward), return code 0."""
    import json
    import math
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSR_BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--mode", "train", "--batch", "64",
                        "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--broadcast-buffers"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and d["value"] > 0
    assert len(d["loss_all_ranks"]) == 2 and all(math.isfinite(v) for v in d["loss_all_ranks"])
    assert d["loss_all_ranks"][0] != d["loss_all_ranks"][1]          # per-rank data shards (seed 42 + rank)
    ddp = d["ddp"]
    assert ddp["world"] == 2 and ddp["broadcast_buffers"] is True and d["comm_wait_ms"] >= 0
    enq = ddp["bucket_enqueue_offsets"]
    assert len(enq) == ddp["buckets"] >= 2 and [e["bucket"] for e in enq] == list(range(len(enq)))
    assert all(e["device_ms"] is not None and e["device_ms"] >= 0 for e in enq)
    assert enq[0]["device_ms"] < enq[-1]["device_ms"]               # buckets leave as backward produces them


def test_bench_infer_two_ranks_fresh_subprocess_gloo():
    """The DEFAULT bench mode (eval forward: the line the driver's N = 1, 2, 4, 8 scaling run reads) with two ranks, started
    as a fresh process like the train test above: replicas sharded by sample, no data-path collective -- only the
    barrier and the MAX-over-ranks of the timed region.  Checks n_gpus == 2, weak scaling, a positive whole-job rate,
    a per-GPU step time, the roofline block, and that no CPU baseline / legs ran at N > 1."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSR_BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]       # whole-job rate = all ranks' samples / time
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert "legs" not in d and "cpu_baseline" not in d
