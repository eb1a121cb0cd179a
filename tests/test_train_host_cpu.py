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


def test_checkpoint_layout_roundtrip(tmp_path):
    import tactilesr_amd
    torch.manual_seed(0)
    m = tactilesr_amd.TactileSR(patternFeatureExtraLayerCnt=1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-2)
    sch = LRWarmupScheduler(torch.optim.lr_scheduler.StepLR(opt, 2, 0.8), epoch_len=10, warmup_t=20,
                            warmup_mode="auto", warmup_factor=1e-4)
    path = os.path.join(tmp_path, "checkpoints", "epoch_3.pth")
    CK.save_checkpoint(path, m, opt, sch, epoch=3)
    ck = torch.load(path, map_location="cpu", weights_only=True)
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
    assert torch.equal(torch.cat([next(iter(s))[0] for s in shards]), LR)
    entries = [[{"LR": LR[i], "depth": torch.zeros(1, 100, 100), "HR": HR[i], "LR_degrade": torch.zeros(1, 4, 4),
                 "alphaBeta": torch.zeros(3)}] for i in range(n)]
    path = os.path.join(tmp_path, "SRdataset_train.npy")
    D.save_dataset(path, entries)
    f = DeviceSRLoader.from_file(path, batch_size=n, device="cpu")
    b = next(iter(f))
    assert torch.equal(b[0], LR) and torch.equal(b[1], HR)
