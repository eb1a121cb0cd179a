"""CPU checks of the host-side training plumbing that sits either side of the hot path:
LR warm-up schedule vs the reference's own sequences, checkpoint layout, dataset file format."""
import os

import numpy as np
import pytest
import torch

from tactilesr_amd.train.lr_scheduler import LRWarmupScheduler
from tactilesr_amd.train import checkpoint as CK


def lr_sequence(cls, cfg, epochs, epoch_len):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=cfg.pop('lr'))
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.pop('step'), gamma=0.8)
    w = cls(sch, epoch_len=epoch_len, **cfg)
    lrs = [opt.param_groups[0]['lr']]
    for _ in range(epochs):
        for _ in range(epoch_len):
            opt.step()
            w.iter_update()
            lrs.append(opt.param_groups[0]['lr'])
        w.epoch_update()
        lrs.append(opt.param_groups[0]['lr'])
    return np.array(lrs), w


CASES = {
    "auto_shipped": dict(lr=1e-3, step=2, by_epoch=True, warmup_t=2000, warmup_by_epoch=False, warmup_mode='auto',
                         warmup_init_lr=1e-5, warmup_factor=1e-4),
    "fix": dict(lr=1e-3, step=2, by_epoch=True, warmup_t=700, warmup_by_epoch=False, warmup_mode='fix',
                warmup_init_lr=1e-5, warmup_factor=1e-4),
    "factor": dict(lr=1e-3, step=1, by_epoch=True, warmup_t=1300, warmup_by_epoch=False, warmup_mode='factor',
                   warmup_init_lr=1e-5, warmup_factor=1e-2),
    "by_epoch": dict(lr=1e-4, step=1, by_epoch=True, warmup_t=3, warmup_by_epoch=True, warmup_mode='auto',
                     warmup_init_lr=1e-5, warmup_factor=1e-1),
    "none": dict(lr=1e-4, step=1, by_epoch=True, warmup_t=0, warmup_by_epoch=False, warmup_mode='fix',
                 warmup_init_lr=0.0, warmup_factor=0.0),
}


@pytest.mark.filterwarnings("ignore:Detected call of")
@pytest.mark.parametrize("name", list(CASES))
def test_lr_warmup_matches_reference_sequence(golden, name):
    ref = golden("lr_schedule")[name]
    got, w = lr_sequence(LRWarmupScheduler, dict(CASES[name]), 6, 500)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()     # same float arithmetic -> bit-equal
    if name == "auto_shipped":   # 'auto' ignores warmup_init_lr: starts at base_lr*factor, ends on StepLR's epoch-4 rate
        assert got[0] == 1e-3 * 1e-4 and abs(got[2003] - 1e-3 * 0.8 ** 2) < 1e-12
    st = w.state_dict()
    assert set(st) == {"n_iter", "n_epoch", "warm_ticks", "torch_scheduler"} and st["n_epoch"] == 6
    assert st["n_iter"] == (0 if CASES[name]["warmup_by_epoch"] else 3000)   # epoch warm-up ignores iter_update


LR_SHORT = dict(lr=1e-4, step=1, by_epoch=True, warmup_t=8, warmup_by_epoch=False, warmup_mode='auto',
                warmup_init_lr=1e-5, warmup_factor=1e-2)


@pytest.mark.filterwarnings("ignore:Detected call of")
def test_lr_warmup_short_schedule_matches_reference_sequence(golden):
    """The short schedule the -m gpu trainer-loop test runs under (warm-up crossing an epoch end, then StepLR(1, 0.8)):
    bit-equal to the reference class's own sequence (tests/golden/lr_schedule_short.npz)."""
    ref = golden("lr_schedule_short")["short_auto"]
    got, _ = lr_sequence(LRWarmupScheduler, dict(LR_SHORT), 3, 6)
    assert np.array_equal(got, ref)


@pytest.mark.filterwarnings("ignore:Detected call of")
@pytest.mark.parametrize("name", ["auto_shipped", "by_epoch"])
@pytest.mark.parametrize("layout", ["own", "reference"])
def test_lr_warmup_resume_mid_schedule_continues_the_reference_sequence(golden, name, layout):
    """Stop after 1.4 epochs (inside the warm-up), save optimizer + scheduler state, rebuild both from scratch, load,
    continue: the concatenated sequence is still the reference's.  `reference`: the state is handed over in the key
    layout the reference's class writes (last_iter / last_epoch / torch_scheduler), as a reference-written checkpoint
    would hold it."""
    ref = golden("lr_schedule")[name]
    cfg = dict(CASES[name])

    def build():
        c = dict(cfg)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=c.pop('lr'))
        sch = torch.optim.lr_scheduler.StepLR(opt, step_size=c.pop('step'), gamma=0.8)
        return opt, LRWarmupScheduler(sch, epoch_len=500, **c)

    def run(opt, w, first_tick, stop_tick, lrs):
        tick = 0
        for _ in range(6):
            for _ in range(500):
                tick += 1
                if first_tick <= tick < stop_tick:
                    opt.step()
                    w.iter_update()
                    lrs.append(opt.param_groups[0]['lr'])
            tick += 1
            if first_tick <= tick < stop_tick:
                w.epoch_update()
                lrs.append(opt.param_groups[0]['lr'])

    opt, w = build()
    lrs = [opt.param_groups[0]['lr']]
    run(opt, w, 1, 701, lrs)                       # 500 iters + epoch end + 199 iters
    so = opt.state_dict()
    ss = w.state_dict() if layout == "own" else w.reference_state_dict()
    opt2, w2 = build()
    opt2.load_state_dict(so)
    w2.load_state_dict(ss)
    run(opt2, w2, 701, 10 ** 9, lrs)
    assert np.array_equal(np.array(lrs), ref)


def test_scheduler_driven_the_way_the_reference_lr_update_hook_does():
    """LRUpdateHook (cpu/hooks/lr_update_hook.py:32-43): after_iter -> iter_update(); after_epoch reads
    `lr_scheduler._is_plateau` and calls epoch_update(metric) for ReduceLROnPlateau, epoch_update() otherwise."""
    import torch
    from tactilesr_amd.train.lr_scheduler import LRWarmupScheduler

    def after_epoch(sch, metric):
        if sch._is_plateau:
            sch.epoch_update(metric)
        else:
            sch.epoch_update()

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5), epoch_len=3)
    assert sch._is_plateau is False
    for _ in range(3):
        opt.step()
        sch.iter_update()
    after_epoch(sch, None)
    assert abs(opt.param_groups[0]["lr"] - 5e-4) < 1e-12
    opt2 = torch.optim.Adam([p], lr=1e-3)
    sch2 = LRWarmupScheduler(torch.optim.lr_scheduler.ReduceLROnPlateau(opt2, factor=0.1, patience=0), epoch_len=3)
    assert sch2._is_plateau is True
    for metric in (1.0, 2.0):            # a worse metric after the first epoch -> lr * 0.1
        for _ in range(3):
            opt2.step()
            sch2.iter_update()
        after_epoch(sch2, metric)
    assert abs(opt2.param_groups[0]["lr"] - 1e-4) < 1e-12


def test_checkpoint_layout_roundtrip(tmp_path):
    import tactilesr_amd
    torch.manual_seed(0)
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-2)
    sch = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt, 2, 0.8), epoch_len=10, warmup_t=20,
                            warmup_mode="auto", warmup_factor=1e-4)
    path = os.path.join(tmp_path, "checkpoints", "epoch_3.pth")
    CK.save_checkpoint(path, m, opt, sch, epoch=3)
    ck = torch.load(path, map_location="cpu", weights_only=False)       # our own file
    assert set(ck) == {"num_gpus", "model", "optimizer", "lr_scheduler", "metric_storage", "epoch"}
    assert ck["num_gpus"] == 1 and ck["epoch"] == 3 and list(ck["model"]) == list(m.state_dict())
    assert os.path.islink(os.path.join(tmp_path, "checkpoints", "latest.pth"))
    torch.manual_seed(1)
    m2 = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    opt2 = torch.optim.Adam(m2.parameters(), lr=5e-2)
    sch2 = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt2, 2, 0.8), epoch_len=10, warmup_t=20,
                             warmup_mode="auto", warmup_factor=1e-4)
    CK.load_checkpoint(os.path.join(tmp_path, "checkpoints", "latest.pth"), m2, opt2, sch2, num_gpus=1)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    with pytest.raises(AssertionError, match="trained with 1 GPUs"):
        CK.load_checkpoint(path, m2, num_gpus=8)


def test_seqs_transplant_replaces_modules_and_freezes_them():
    """train/tactileSRSeqs_train.py:43-59,74-77: the optimizer is built BEFORE the transplant, so it keeps the
    discarded modules' parameters."""
    import tactilesr_amd
    single = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    seqs = tactilesr_amd.TactileSR(seqsCnt=7, patternFeatureExtraLayerCnt=1)
    opt = torch.optim.Adam(seqs.parameters(), lr=1e-4)
    sd = {k: v.clone() for k, v in single.state_dict().items()}
    CK.model_param_init(seqs, sd, lambda: tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1))
    w = seqs.patternFeatureExtra_layer[0].conv_3_1[0].weight
    assert torch.equal(w, sd["patternFeatureExtra_layer.0.conv_3_1.0.weight"])
    opt_ids = {id(p) for g in opt.param_groups for p in g["params"]}
    assert id(w) not in opt_ids                       # frozen: not among the optimizer's parameters
    assert id(seqs.inputContact_layer[0].weight) in opt_ids
    n_before = len(tactilesr_amd.TactileSR(seqsCnt=7, patternFeatureExtraLayerCnt=1).state_dict())
    assert len(seqs.state_dict()) == n_before          # key set unchanged by the transplant


def test_dataset_file_format_reads_like_the_reference_loader(tmp_path):
    from tactilesr_amd.data import depth2tactile as D
    entries = [[{"LR": torch.full((3, 4, 4), float(i)), "depth": torch.zeros(1, 100, 100),
                 "HR": torch.ones(1, 100, 100) * i, "LR_degrade": torch.zeros(1, 4, 4),
                 "alphaBeta": torch.tensor([1.0, 2.0, 3.0])}] for i in range(5)]
    path = os.path.join(tmp_path, "SRdataset_train.npy")
    D.save_dataset(path, entries)
    ds = np.load(path, allow_pickle=True)            # our own file: utility/load_tactile_dataset.py:41
    assert len(ds) == 5
    item = ds[3].item()                              # :44  self.SRdataset[idx].item()['LR']
    assert set(item) == {"LR", "depth", "HR", "LR_degrade", "alphaBeta"}
    assert np.ascontiguousarray(item["LR"]).shape == (3, 4, 4) and float(item["HR"].mean()) == 3.0


def test_device_resident_loader_protocol(tmp_path):
    """DeviceSRLoader yields the (LR, HR) tuples the trainer glue takes from the reference's DataLoader: sequential
    order without shuffle, a fresh seeded permutation per epoch with it, drop_last, rank shards, and it reads the
    generator's dataset file."""
    from tactilesr_amd.data import DeviceSRLoader, depth2tactile as D
    n = 11
    LR = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 3, 4, 4).clone()
    HR = LR[:, :1, :1, :1].expand(n, 1, 100, 100).clone() * 2
    ld = DeviceSRLoader(LR, HR, batch_size=4, device="cpu")
    assert len(ld) == 3
    got = [b for b in ld]
    assert [b[0].shape[0] for b in got] == [4, 4, 3]
    assert torch.equal(torch.cat([b[0] for b in got]), LR) and torch.equal(torch.cat([b[1] for b in got]), HR)
    ld = DeviceSRLoader(LR, HR, batch_size=4, shuffle=True, drop_last=True, seed=7, device="cpu")
    assert len(ld) == 2
    e1 = torch.cat([b[0][:, 0, 0, 0] for b in ld])
    e2 = torch.cat([b[0][:, 0, 0, 0] for b in ld])
    assert e1.numel() == 8 and len(set(e1.tolist())) == 8 and not torch.equal(e1, e2)
    again = DeviceSRLoader(LR, HR, batch_size=4, shuffle=True, drop_last=True, seed=7, device="cpu")
    assert torch.equal(torch.cat([b[0][:, 0, 0, 0] for b in again]), e1)
    for b in DeviceSRLoader(LR, HR, batch_size=4, shuffle=True, seed=1, device="cpu"):   # pairs stay aligned
        assert torch.equal(b[1][:, 0, 0, 0], b[0][:, 0, 0, 0] * 2)
    shards = [DeviceSRLoader(LR, HR, batch_size=16, device="cpu", rank=r, world_size=2) for r in range(2)]
    got = [next(iter(s))[0][:, 0, 0, 0].tolist() for s in shards]          # equal sizes; the tail wraps to the head
    assert got == [[0, 1, 2, 3, 4, 5], [6, 7, 8, 9, 10, 0]]
    entries = [[{"LR": LR[i], "depth": torch.zeros(1, 100, 100), "HR": HR[i], "LR_degrade": torch.zeros(1, 4, 4),
                 "alphaBeta": torch.zeros(3)}] for i in range(n)]
    path = os.path.join(tmp_path, "SRdataset_train.npy")
    D.save_dataset(path, entries)
    f = DeviceSRLoader.from_file(path, batch_size=n, device="cpu")
    b = next(iter(f))
    assert torch.equal(b[0], LR) and torch.equal(b[1], HR)


# ---------------------------------------------------------------------------------------------------------------
# checkpoint interop with the reference's trainer (cpu/trainer.py:394-498), metric store, config shim
# ---------------------------------------------------------------------------------------------------------------
class _ForeignMetricStore(dict):
    """Stand-in for the reference's pickled MetricStorage instance (an arbitrary python object under
    'metric_storage': what makes every reference-written checkpoint need a trusted load)."""


class _Taker:
    def __init__(self):
        self.state = None

    def load_state_dict(self, st):
        self.state = st


@pytest.mark.filterwarnings("ignore:Detected call of")
def test_load_checkpoint_laid_out_as_the_reference_writes_it(tmp_path, golden):
    """A dict with exactly the reference's key set -- 'lr_scheduler' and 'optimizer' are the REFERENCE's own objects'
    states 700 ticks into the shipped schedule (tests/golden/ref_lr_state.pth, written by the reference's class),
    'metric_storage' a pickled object, plus 'hooks' and 'grad_scaler' -- resumes here: the LR sequence continues the
    reference's golden sequence, hook and scaler states reach their takers, start_iter follows :452-457."""
    import tactilesr_amd
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_lr_state.pth"), weights_only=True)
    torch.manual_seed(0)
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    scaler_state = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                    "_growth_tracker": 7}
    ck = {"num_gpus": 1, "model": m.state_dict(), "optimizer": fx["optimizer"], "lr_scheduler": fx["lr_scheduler"],
          "metric_storage": _ForeignMetricStore(total_loss=[0.5]), "epoch": 0,
          "hooks": {"EvalHook": {"best": 1.0}, "UnknownHook": {}}, "grad_scaler": scaler_state}
    path = os.path.join(tmp_path, "epoch_0.pth")
    torch.save(ck, path)
    torch.manual_seed(1)
    m2 = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    cfg = dict(CASES["auto_shipped"])
    cfg.pop("lr")
    sch = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt, cfg.pop("step"), 0.8), epoch_len=500, **cfg)
    with pytest.raises(Exception):
        CK.load_checkpoint(path, m2, opt, sch, grad_scaler=_Taker())            # pickled object: needs trusted=True
    with pytest.raises(AssertionError, match="inconsistent AMP"):
        CK.load_checkpoint(path, m2, opt, sch, trusted=True)                    # file has grad_scaler, caller has none
    scaler, evalhook, ckpthook = _Taker(), _Taker(), _Taker()
    got = CK.load_checkpoint(path, m2, opt, sch, num_gpus=1, trusted=True, grad_scaler=scaler,
                             hooks={"EvalHook": evalhook, "CheckpointHook": ckpthook}, epoch_len=500)
    assert got["start_iter"] == 500 and isinstance(got["metric_storage"], _ForeignMetricStore)
    assert scaler.state == scaler_state and evalhook.state == {"best": 1.0} and ckpthook.state is None
    assert got["hooks_missing"] == ["CheckpointHook"] and got["hooks_unexpected"] == ["UnknownHook"]
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # continue from tick 700 (699 iterations + 1 epoch end) to the end of the 6-epoch fixture
    ref = golden("lr_schedule")["auto_shipped"]
    lrs, tick = [], 0
    for _ in range(6):
        for _ in range(500):
            tick += 1
            if tick >= 701:
                opt.step()
                sch.iter_update()
                lrs.append(opt.param_groups[0]["lr"])
        tick += 1
        if tick >= 701:
            sch.epoch_update()
            lrs.append(opt.param_groups[0]["lr"])
    assert np.array_equal(np.array(lrs), ref[701:])
    assert opt.param_groups[0]["lr"] == ref[-1]


@pytest.mark.filterwarnings("ignore:Detected call of")
def test_reference_state_dict_matches_the_reference_class_at_the_same_tick():
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_lr_state.pth"), weights_only=True)
    cfg = dict(CASES["auto_shipped"])
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=cfg.pop("lr"))
    sch = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt, cfg.pop("step"), 0.8), epoch_len=500, **cfg)
    for _ in range(500):
        opt.step()
        sch.iter_update()
    sch.epoch_update()
    for _ in range(199):
        opt.step()
        sch.iter_update()
    mine, ref = sch.reference_state_dict(), fx["lr_scheduler"]
    assert set(mine) <= set(ref)
    for k in ("last_iter", "last_epoch", "in_iter_warmup"):
        assert mine[k] == ref[k], k
    for k in ("last_epoch", "_step_count", "_last_lr"):
        assert mine["torch_scheduler"][k] == ref["torch_scheduler"][k], k
    assert opt.param_groups[0]["lr"] == fx["optimizer"]["param_groups"][0]["lr"]


def test_saved_checkpoint_carries_a_metric_store_the_reference_protocol_accepts(tmp_path):
    """save_checkpoint's default 'metric_storage' is an object the reference's hooks can keep using after installing
    it as the live store (cpu/trainer.py:469; cpu/hooks/logger_hook.py:38-95; cpu/hooks/lr_update_hook.py:29-37)."""
    import tactilesr_amd
    from tactilesr_amd.train.metrics import MetricStorage
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    ms = MetricStorage(window_size=3)
    for it, v in enumerate([4.0, 2.0, 6.0, 8.0]):
        ms.update(iter=it, total_loss=v)
        ms.update(iter=it, smooth=False, lr=1e-3 * (it + 1))
    path = os.path.join(tmp_path, "c", "iter_3.pth")
    CK.save_checkpoint(path, m, iteration=3, metric_storage=ms)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)
    ck = CK.load_checkpoint(path, m)                 # own classes: restricted unpickler is enough
    st = ck["metric_storage"]
    assert ck["start_iter"] == 4 and isinstance(st, MetricStorage) and set(st) == {"total_loss", "lr"}
    assert st["total_loss"].latest == 8.0 and abs(st["total_loss"].avg - 16.0 / 3) < 1e-12      # window of 3
    assert st["total_loss"].global_avg == 5.0 and st["total_loss"].global_sum == 20.0
    assert st.values_maybe_smooth == {"total_loss": (3, 16.0 / 3), "lr": (3, 4e-3)}
    st.update(iter=4, total_loss=1.0)                                    # keeps working as the live store
    assert st["total_loss"].latest == 1.0 and "Eval Metric" not in st
    with pytest.raises(AssertionError):
        st.update(iter=4, total_loss=1.0)                                # iterations must increase
    with pytest.raises(AssertionError):
        st.update(iter=5, smooth=False, total_loss=1.0)                  # smooth flag is fixed per metric
    CK.save_checkpoint(os.path.join(tmp_path, "c", "epoch_0.pth"), m, epoch=0)             # default: empty store
    assert isinstance(CK.load_checkpoint(os.path.join(tmp_path, "c", "latest.pth"), m, trusted=True)["metric_storage"],
                      MetricStorage)


def test_config_default_carries_the_reference_keys_and_values():
    """tactilesr_amd.config.default vs the reference's three dicts (tests/golden/config_default.json, extracted from
    config/default.py:8-96): same key sets, same values; paths hang off root_path instead of /code; importing it
    never touches a GPU and 'device' resolves lazily."""
    import json
    from tactilesr_amd.config import default as D
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_default.json")))
    for name, rd in ref.items():
        mine = getattr(D, name)
        assert set(mine) == set(rd), (name, set(mine) ^ set(rd))
        for k, v in rd.items():
            if isinstance(v, str) and v.startswith("<root>"):
                assert mine[k] == os.path.join(D.root_path, v[len("<root>/"):]) if len(v) > 6 else mine[k] == D.root_path
            else:
                assert mine[k] == v and type(mine[k]) is type(v), (name, k, mine[k], v)
    assert D.tactileSeqs_config["seqsCnt"] == 7 and D.tactileSR_config["warmup_by_epoch"] is True
    if not torch.cuda.is_available():
        from tactilesr_amd._lib import TactileSRHipError
        with pytest.raises(TactileSRHipError):
            D.device


def test_seqs_index_table_follows_the_reference_arithmetic():
    """data/SeqsDataset/seqsDepth2Tactile.py:50-56: taps 0..25 degrees use the LAST sample of their sequence, the
    30-degree tap uses seqs_idx; contact stride is 81 sequences; column order is newest (30 degrees) first."""
    from tactilesr_amd.data.seqs_depth2tactile import seqs_index_table
    for nc, nt, sc in ((2, 3, 4), (18, 9, 16)):
        idx, trans = seqs_index_table(nc, nt, sc)
        k = 0
        for c in range(nc):
            for t in range(nt):
                for s in range(sc):
                    exp = [s + sc * (6 + t * 9) + sc * 81 * c] + \
                          [sc - 1 + sc * (r + t * 9) + sc * 81 * c for r in (5, 4, 3, 2, 1, 0)]
                    assert idx[k].tolist() == exp and int(trans[k]) == t
                    k += 1
        assert k == idx.shape[0]
    assert seqs_index_table()[0].shape == (18 * 9 * 16, 7)           # the shipped generator: 2592 items
