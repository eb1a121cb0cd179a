"""world_size-2 gloo tests (CPU) of the data-parallel plumbing (tactilesr_amd.ddp): the gradient arena in
backward-production order, bucket all-reduces enqueued from INSIDE backward, the late path of the first step and of
gradient accumulation, parameter + buffer broadcast, equal-size data shards.  The HIP engine cannot run here, so a toy
autograd function plays its part through the very same GradSink / GradArena / GradSync objects the engine uses."""
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


class _ToyEngine:
    """Stands in for model/_train.py's TrainEngine: y = sum_i relu-free linear pieces; backward produces the
    parameter gradients in REVERSE layer order through a GradSink, like the HIP engine does."""

    def __init__(self, module):
        self.m = module
        self.arena = None
        self.grad_sync = None
        self.n_buckets = 3
        self.log = []                # ("backward_end",) markers, interleaved with the GradSync's events by the test

    def backward(self, x, dy, token=None):
        from tactilesr_amd.ddp import GradSink
        named = dict(self.m.named_parameters())
        sink = GradSink(self, named, x.device, token=token)
        # y = (x @ W2^T + b2) summed with (x @ W1^T + b1): two independent linear layers, "layer 2" finishes first
        for name in ("l2", "l1"):
            lin = getattr(self.m, name)
            gw = sink.dest(f"{name}.weight", lin.weight.shape)
            torch.matmul(dy.t(), x, out=gw)
            sink.put(f"{name}.weight", gw)
            sink.put_copy(f"{name}.bias", dy.sum(0))
        sink.put_copy("scale", (dy * 0).sum().reshape(1) + 1.0)          # a third bucket's worth
        out = sink.finalize()
        if self.grad_sync is not None:
            self.grad_sync.events.append(("backward_end", -1))
        return out


class _Token:
    """Stands in for the engine's per-forward context object (weakly referenced by ddp.note_forward)."""


class _ToyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, names, x, *params):
        m = engine.m
        ctx.engine, ctx.names, ctx.x = engine, names, x
        ctx.token = _Token()
        if any(ctx.needs_input_grad):
            from tactilesr_amd.ddp import note_forward
            note_forward(engine, ctx.token)            # like TactileSRTrainFn / BlockTrainFn do with their context
        return x @ m.l1.weight.t() + m.l1.bias + x @ m.l2.weight.t() + m.l2.bias + m.scale * 0

    @staticmethod
    def backward(ctx, dy):
        g = ctx.engine.backward(ctx.x, dy.contiguous(), ctx.token)
        return (None, None, None) + tuple(g[n] for n in ctx.names)


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l1 = torch.nn.Linear(40, 24)
        self.l2 = torch.nn.Linear(40, 24)
        self.scale = torch.nn.Parameter(torch.ones(1))
        self.bn = torch.nn.BatchNorm1d(24)             # only for its buffers (broadcast test)
        self.frozen = torch.nn.Parameter(torch.randn(5), requires_grad=False)
        self._engine = _ToyEngine(self)

    def train_engine(self):
        return self._engine

    def forward(self, x):
        named = [(n, p) for n, p in self.named_parameters() if n in ("l1.weight", "l1.bias", "l2.weight", "l2.bias",
                                                                     "scale")]
        return _ToyFn.apply(self._engine, [n for n, _ in named], x, *[p for _, p in named])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tactilesr_amd import ddp
    r, w, _ = ddp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                  # replicas start different on purpose
    model = _Toy()
    with torch.no_grad():
        model.bn.running_mean.fill_(float(rank + 1))
        model.bn.num_batches_tracked.fill_(7 * (rank + 1))
    sync = ddp.GradSync(model)
    assert model.train_engine().grad_sync is sync
    sync.broadcast_parameters(0)
    state = torch.cat([t.detach().flatten().double() for t in list(model.parameters()) + list(model.buffers())])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 40, generator=g)
    a, b = ddp.shard_batch(6, rank, world)
    res = {"rank": rank, "state_sum": float(state.sum()), "state_abs": float(state.abs().sum()),
           "frozen0": float(model.frozen[0]), "bn_mean": float(model.bn.running_mean[0]),
           "nbt": int(model.bn.num_batches_tracked)}

    def step(zero_to_none=True):
        for p in model.parameters():
            if zero_to_none:
                p.grad = None
        sync.events.clear()
        loss = model(x[a:b]).pow(2).mean()
        loss.backward()
        local = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        ev_before_finish = list(sync.events)
        sync.finish()
        return local, ev_before_finish

    def expect_mean(local):
        flat = torch.cat([local[n].flatten() for n in sorted(local)])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        return sum(gathered) / world

    # step 1: no arena yet -> built at the end of backward in production order; buckets reduced late in finish()
    local, ev = step()
    arena = model.train_engine().arena
    res["order"] = list(arena.names)
    res["first_step_events"] = ev
    got = torch.cat([dict(model.named_parameters())[n].grad.flatten() for n in sorted(local)])
    res["step1_ok"] = bool(torch.allclose(got, expect_mean(local), atol=1e-6))
    res["buckets"] = list(arena.buckets)
    res["late1"] = [e for e in sync.events if e[0] == "enqueue_late"]
    # step 2: arena exists, grads None -> DIRECT: every bucket is enqueued before backward returns.  NB the local
    # gradients were cloned after backward returned, i.e. while the all-reduce may already have run: recompute them
    for p in model.parameters():
        p.grad = None
    sync.events.clear()
    loss = model(x[a:b]).pow(2).mean()
    loss.backward()
    res["direct_events"] = list(sync.events)
    sync.finish()
    named = dict(model.named_parameters())
    res["grads_in_arena"] = all(named[n].grad.data_ptr() == arena.flat.data_ptr() + 4 * arena.offsets[n]
                                for n in arena.names)
    # reference for step 2: the same maths without any sync machinery
    ref_model = _Toy()
    ref_model.load_state_dict(model.state_dict())
    ref_model._engine.grad_sync = None
    ref_model(x[a:b]).pow(2).mean().backward()
    local2 = {n: p.grad.detach().clone() for n, p in ref_model.named_parameters() if p.grad is not None}
    got2 = torch.cat([named[n].grad.flatten() for n in sorted(local2)])
    res["step2_ok"] = bool(torch.allclose(got2, expect_mean(local2), atol=1e-6))
    # step 3: gradient accumulation (grads NOT reset): nothing may be issued early; result = mean of (g2_mean + g3_local)
    prev = {n: named[n].grad.detach().clone() for n in local2}
    sync.events.clear()
    model(x[a:b]).pow(2).mean().backward()
    res["accum_events"] = list(sync.events)
    acc_local = {n: named[n].grad.detach().clone() for n in local2}
    sync.finish()
    got3 = torch.cat([named[n].grad.flatten() for n in sorted(local2)])
    res["step3_ok"] = bool(torch.allclose(got3, expect_mean(acc_local), atol=1e-6))
    res["accum_is_sum"] = bool(all(torch.allclose(acc_local[n], prev[n] + local2[n], atol=1e-6) for n in local2))
    st = sync.stats()
    res["stats_ok"] = (st["world"] == world and st["buckets"] == 3 and st["comm_wait_ms"] >= 0
                       and [e["bucket"] for e in st["bucket_enqueue_offsets"]] == [0, 1, 2]
                       and all(e["host_ms"] >= 0 for e in st["bucket_enqueue_offsets"]))
    # step 4: the USUAL accumulation loop -- zero_grad(); backward (micro-batch 1, under no_sync); backward (micro-batch
    # 2); finish() -- must give mean over ranks of (g1 + g2); nothing may leave before the last micro-batch
    xs = [x[a:b], x[a:b] * 0.5 + 1.0]
    refs = []
    for xm in xs:
        rm = _Toy()
        rm.load_state_dict(model.state_dict())
        rm._engine.grad_sync = None
        rm(xm).pow(2).mean().backward()
        refs.append({n: p.grad.detach().clone() for n, p in rm.named_parameters() if p.grad is not None})
    want = {n: refs[0][n] + refs[1][n] for n in refs[0]}
    for p in model.parameters():
        p.grad = None
    sync.events.clear()
    with sync.no_sync():
        model(xs[0]).pow(2).mean().backward()
    res["nosync_events"] = [e for e in sync.events if e[0].startswith("enqueue")]
    model(xs[1]).pow(2).mean().backward()
    sync.finish()
    got4 = torch.cat([named[n].grad.flatten() for n in sorted(want)])
    res["step4_ok"] = bool(torch.allclose(got4, expect_mean(want), atol=1e-6))
    # step 5: the same loop WITHOUT no_sync: micro-batch 1 puts its buckets on the wire from inside backward, so the
    # second backward must refuse to add local gradients on top of them
    for p in model.parameters():
        p.grad = None
    model(xs[0]).pow(2).mean().backward()
    try:
        model(xs[1]).pow(2).mean().backward()
        res["step5_raised"] = False
    except RuntimeError as e:
        res["step5_raised"] = "no_sync" in str(e)
    sync.finish()                       # drain the collectives of micro-batch 1 on both ranks
    # step 6: the engine applied TWICE inside one graph (f(m(a)) + f(m(b))): autograd sums the two applications'
    # gradients only after both backward passes ran, so neither may write the arena slots; nothing leaves from inside
    # backward, finish() reduces the summed gradients: mean over ranks of (g(a) + g(b))
    for p in model.parameters():
        p.grad = None
    sync.events.clear()
    (model(xs[0]).pow(2).mean() + model(xs[1]).pow(2).mean()).backward()
    res["shared_events"] = [e for e in sync.events if e[0].startswith("enqueue")]
    sync.finish()
    got6 = torch.cat([named[n].grad.flatten() for n in sorted(want)])
    res["step6_ok"] = bool(torch.allclose(got6, expect_mean(want), atol=1e-6))
    # ... and the step after it is a plain direct step again
    for p in model.parameters():
        p.grad = None
    sync.events.clear()
    model(xs[0]).pow(2).mean().backward()
    res["after_shared_events"] = [e for e in sync.events if e[0] == "enqueue"]
    sync.finish()
    got7 = torch.cat([named[n].grad.flatten() for n in sorted(want)])
    res["step7_ok"] = bool(torch.allclose(got7, expect_mean(refs[0]), atol=1e-6))
    # broadcast_buffers=True (torch-DDP default): rank 0's BatchNorm statistics before every forward
    with torch.no_grad():
        model.bn.running_mean.fill_(10.0 + rank)
        model.bn.running_var.fill_(2.0 + rank)
        model.bn.num_batches_tracked.fill_(3 + rank)
    sync.pre_forward()                  # broadcast_buffers=False: stays rank-local
    res["bb_off"] = (float(model.bn.running_mean[0]), int(model.bn.num_batches_tracked))
    sync.broadcast_buffers = True
    sync.pre_forward()
    res["bb_on"] = (float(model.bn.running_mean[0]), float(model.bn.running_var[0]), int(model.bn.num_batches_tracked),
                    sync.buffer_broadcasts)
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_gloo_world2_overlapped_buckets_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    # broadcast: parameters (frozen included) AND buffers identical on both ranks, = rank 0's
    assert r0["state_sum"] == r1["state_sum"] and r0["state_abs"] == r1["state_abs"]
    assert r0["frozen0"] == r1["frozen0"] and r1["bn_mean"] == 1.0 and r1["nbt"] == 7
    for r in res:
        assert r["step1_ok"] and r["step2_ok"] and r["step3_ok"] and r["accum_is_sum"] and r["grads_in_arena"], r
        # the arena follows the order backward produced the gradients in
        assert r["order"] == ["l2.weight", "l2.bias", "l1.weight", "l1.bias", "scale"]
        b = r["buckets"]
        assert len(b) == 3 and b[0][0] == 0 and all(b[i][1] == b[i + 1][0] for i in range(2))
        # first step: nothing could be issued early (no arena yet) -> all buckets late
        assert [e[0] for e in r["first_step_events"]] == ["backward_end"] and len(r["late1"]) == 3
        # direct step: ALL bucket all-reduces were enqueued BEFORE backward returned, in production order
        ev = r["direct_events"]
        assert ev == [("enqueue", 0), ("enqueue", 1), ("enqueue", 2), ("backward_end", -1)], ev
        # accumulation step: nothing issued from inside backward
        assert [e[0] for e in r["accum_events"]] == ["backward_end"]
        assert r["stats_ok"]
        # zero_grad; backward under no_sync; backward; finish == mean(g1 + g2), nothing on the wire during micro-batch 1
        assert r["step4_ok"] and r["nosync_events"] == []
        assert r["step5_raised"] is True
        assert r["step6_ok"] and r["shared_events"] == [], r
        assert r["step7_ok"] and r["after_shared_events"] == [("enqueue", 0), ("enqueue", 1), ("enqueue", 2)], r
        assert r["bb_off"] == (10.0 + r["rank"], 3 + r["rank"]) and r["bb_on"] == (10.0, 2.0, 3, 1)


def test_engine_applied_twice_in_one_graph_sums_both_gradients():
    """ADVICE r03 (ddp.py:133): with the arena built, `(f(m(x1)) + f(m(x2))).backward()` used to hand the SAME arena
    slots to both backward passes -- the second overwrote the first's views before autograd summed them (l1.weight off by
    ~1 max-abs).  Every application of a multiply-applied engine now returns fresh tensors; a graph dropped without a
    backward does not leave the engine stuck in that mode; an un-differentiated forward (no_grad) does not count."""
    torch.manual_seed(0)
    m = _Toy()
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(5, 40, generator=g), torch.randn(7, 40, generator=g)
    names = ("l1.weight", "l1.bias", "l2.weight", "l2.bias")

    def separate():
        out = {}
        for x in (x1, x2):
            r = _Toy()
            r.load_state_dict(m.state_dict())
            r(x).pow(2).mean().backward()
            for n, p in r.named_parameters():
                if n in names:
                    out[n] = out.get(n, 0) + p.grad
        return out
    want = separate()
    named = dict(m.named_parameters())
    for attempt in range(2):                             # no arena yet: a shared graph never lays it out
        for p in m.parameters():
            p.grad = None
        (m(x1).pow(2).mean() + m(x2).pow(2).mean()).backward()
        for n in names:
            assert torch.allclose(named[n].grad, want[n], atol=1e-6), (attempt, n)
        assert m.train_engine().arena is None
    # a plain step builds the arena; then the shared graph again, now WITH an arena to (not) overwrite
    for p in m.parameters():
        p.grad = None
    m(x1).pow(2).mean().backward()
    arena = m.train_engine().arena
    assert arena is not None
    for p in m.parameters():
        p.grad = None
    (m(x1).pow(2).mean() + m(x2).pow(2).mean()).backward()
    for n in names:
        assert torch.allclose(named[n].grad, want[n], atol=1e-6), n
    # a forward whose graph is dropped, and one under no_grad, leave nothing behind: the next step is direct again
    y = m(x2)
    del y
    with torch.no_grad():
        m(x2)
    for p in m.parameters():
        p.grad = None
    m(x1).pow(2).mean().backward()
    assert all(named[n].grad.data_ptr() == arena.flat.data_ptr() + 4 * arena.offsets[n] for n in arena.names)


def test_shard_batch_covers_everything():
    from tactilesr_amd import ddp
    for n in (1, 7, 8, 65536):
        for world in (1, 2, 8):
            spans = [ddp.shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_equal_shards_give_every_rank_the_same_batch_count():
    """n not divisible by world*batch: every rank must still run the same number of batches per epoch (a rank that
    stops early leaves the others blocked in their all-reduce), and every sample must be seen."""
    from tactilesr_amd.data import DeviceSRLoader
    n, world, bs = 100, 8, 4
    LR = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 3, 4, 4).clone()
    HR = torch.zeros(n, 1, 2, 2)
    loaders = [DeviceSRLoader(LR, HR, batch_size=bs, device="cpu", rank=r, world_size=world) for r in range(world)]
    assert len({len(ld) for ld in loaders}) == 1 and len(loaders[0]) == 4          # ceil(ceil(100/8)/4)
    seen = set()
    for ld in loaders:
        sizes = [b[0].shape[0] for b in ld]
        assert sizes == [4, 4, 4, 1]
        for b in ld:
            seen.update(int(v) for v in b[0][:, 0, 0, 0])
    assert seen == set(range(n))
