#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; it imports
model/tactileSR_model.py, model/tPSFNet.py and utility/tools.py from there, CPU
only).  Nothing from the reference is copied: the fixtures hold inputs, seeds and
the reference's outputs.  Parameters are not stored -- they are regenerated from a
seed by ``oracle.tactilesr_oracle.random_state_dict`` (deterministic CPU RNG) and a
sha256 of their bytes is stored to detect drift.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import copy
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from model.tactileSR_model import TactileSR  # noqa: E402  (reference)
from model.tPSFNet import tPSFNet  # noqa: E402  (reference)
from utility.tools import calculationPSNR, calculationSSIM  # noqa: E402  (reference)

from oracle import tactilesr_oracle as O  # noqa: E402

torch.set_num_threads(8)


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def probe(t):
    """Small deterministic sub-sample + global checksums of a (B,C,H,W) tensor."""
    t = t.detach()
    cs = max(1, t.shape[1] // 4)
    return {
        "probe": t[:, ::cs, ::3, ::3].contiguous().numpy(),
        "sum": np.float64(t.double().sum().item()),
        "abssum": np.float64(t.double().abs().sum().item()),
        "absmax": np.float64(t.abs().max().item()),
    }


def flat(prefix, d, out):
    for k, v in d.items():
        out[f"{prefix}/{k}"] = v


def load_random(model, seed, **cfg):
    shapes = O.tactilesr_state_shapes(**cfg)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict key order mismatch"
    for k in shapes:
        assert tuple(ref_sd[k].shape) == tuple(shapes[k]), k
    sd = O.random_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    return sd


def stage_hooks(model, stages):
    hs = []

    def add(name, mod):
        hs.append(mod.register_forward_hook(lambda m, i, o, name=name: stages.__setitem__(name, o.detach().clone())))

    for t, m in enumerate(model.inputLayer_pattern_list):
        add(f"stem{t}", m)
    add("fuse", model.inputContact_layer)
    for i, m in enumerate(model.patternFeatureExtra_layer):
        add(f"msrb{i}", m)
    add("force_in", model.input_layer_force)
    add("force", model.forceFeatureExtra_layer)
    add("head0", model.output_layer[1])
    return hs


def gen_init():
    """F1: seeded construction -> state_dict hash + probes (init RNG-order parity)."""
    out = {}
    for tag, cfg in (("t1", dict()), ("t7", dict(seqsCnt=7))):
        torch.manual_seed(42)
        m = TactileSR(**cfg)
        sd = m.state_dict()
        out[f"{tag}/sha256"] = np.array(sd_hash(sd))
        out[f"{tag}/nkeys"] = np.int64(len(sd))
        out[f"{tag}/nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
        for k in ("patternFeatureExtra_layer.0.conv_3_1.0.weight", "patternFeatureExtra_layer.5.confusion.bias",
                  "forceFeatureExtra_layer.0.conv2.bias", "inputLayer_pattern_list.0.1.weight",
                  "output_layer.2.weight", "input_layer_force.1.weight"):
            out[f"{tag}/probe/{k}"] = sd[k].flatten()[:32].numpy()
    torch.manual_seed(42)
    net = tPSFNet(gama=1.4, perception_scale=None, device="cpu")
    sd = net.state_dict()
    out["tpsf/sha256"] = np.array(sd_hash(sd))
    out["tpsf/nparams"] = np.int64(sum(p.numel() for p in net.parameters()))
    out["tpsf/probe/MLP_layer.1.weight"] = sd["MLP_layer.1.weight"].flatten()[:32].numpy()
    out["tpsf/probe/MLP_layer.7.bias"] = sd["MLP_layer.7.bias"].numpy()
    np.savez_compressed(os.path.join(HERE, "init.npz"), **out)


def gen_eval():
    """F2/F4/F5: eval-mode forward, randomised parameters."""
    out = {}
    for tag, cfg, B, seed in (("t1", dict(), 2, 101), ("t7", dict(seqsCnt=7), 2, 107),
                              ("sf25t8", dict(scale_factor=25, seqsCnt=8), 1, 125),
                              ("t1_l2", dict(patternFeatureExtraLayerCnt=2), 3, 131)):
        m = TactileSR(**cfg)
        sd = load_random(m, seed, **cfg)
        m.eval()
        g = torch.Generator().manual_seed(seed + 1)
        T = cfg.get("seqsCnt", 1)
        LR = torch.rand(B, 3 * T, 4, 4, generator=g) * 8
        stages = {}
        hs = stage_hooks(m, stages)
        with torch.no_grad():
            y = m(LR)
        for h in hs:
            h.remove()
        # fp64 run of the same reference module: the conditioning yardstick (how far the
        # reference's own fp32 CPU result is from exact arithmetic on these inputs)
        with torch.no_grad():
            y64 = m.double()(LR.double())
        m.float()
        out[f"{tag}/ref32_vs_f64"] = np.float64(((y.double() - y64).abs().max() / y64.abs().max()).item())
        out[f"{tag}/seed"] = np.int64(seed)
        out[f"{tag}/sha256"] = np.array(sd_hash(sd))
        out[f"{tag}/LR"] = LR.numpy()
        if tag == "sf25t8":
            flat(f"{tag}/out", probe(y), out)
            out[f"{tag}/out_full0"] = y[0, 0, ::2, ::2].numpy()
            out[f"{tag}/out64_full0"] = y64[0, 0, ::2, ::2].numpy()
        else:
            out[f"{tag}/out"] = y.numpy()
            out[f"{tag}/out64"] = y64.numpy()
        for name, t in stages.items():
            flat(f"{tag}/stage/{name}", probe(t), out)
        print(tag, "out", tuple(y.shape), float(y.abs().max()), "ref32_vs_f64", out[f"{tag}/ref32_vs_f64"])
    np.savez_compressed(os.path.join(HERE, "eval.npz"), **out)


def gen_eval_init():
    """configs[4] shape (scale_factor=25, seqsCnt=8) with the reference's OWN parameters: `_init_network` under
    torch.manual_seed(42) (model/tactileSR_model.py:92-98; config/default.py:10), BatchNorm running statistics moved
    off (0, 1) by two train-mode passes of the reference itself, then an eval forward in fp32 and fp64.  The fixture
    stores the input, the moved running statistics (the weights are reproduced from the seed on the test side: the
    drop-in's constructor is bit-identical, tests/test_host_cpu.py) and the outputs.  Same for the shipped shape
    (sf=10, T=1) as `init_t1`."""
    out = {}
    for tag, cfg, B in (("init_sf25t8", dict(scale_factor=25, seqsCnt=8), 2), ("init_t1", dict(), 4)):
        torch.manual_seed(42)
        m = TactileSR(**cfg)
        T, sf = cfg.get("seqsCnt", 1), cfg.get("scale_factor", 10)
        out[f"{tag}/sha256_init"] = np.array(sd_hash(m.state_dict()))
        g = torch.Generator().manual_seed(4242)
        m.train()
        with torch.no_grad():
            for _ in range(2):
                m(torch.rand(B, 3 * T, 4, 4, generator=g) * 8)
        m.eval()
        sd = m.state_dict()
        for k, v in sd.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out[f"{tag}/stat/{k}"] = v.numpy().copy()
        LR = torch.rand(B, 3 * T, 4, 4, generator=g) * 8
        stages = {}
        hs = stage_hooks(m, stages)
        with torch.no_grad():
            y = m(LR)
        for h in hs:
            h.remove()
        with torch.no_grad():
            y64 = m.double()(LR.double())
        m.float()
        out[f"{tag}/ref32_vs_f64"] = np.float64(((y.double() - y64).abs().max() / y64.abs().max()).item())
        out[f"{tag}/LR"] = LR.numpy()
        out[f"{tag}/out"] = y.numpy()
        out[f"{tag}/out64"] = y64.numpy()
        for name, t in stages.items():
            flat(f"{tag}/stage/{name}", probe(t), out)
        print(tag, "out", tuple(y.shape), float(y.abs().max()), "ref32_vs_f64", out[f"{tag}/ref32_vs_f64"])
    np.savez_compressed(os.path.join(HERE, "eval_init.npz"), **out)


def gen_train():
    """F3: train-mode fwd + bwd + one Adam(L2) step with the reference's own step
    semantics (train/tactileSR_train.py:41-51, cpu/trainer.py:346-362)."""
    out = {}
    keys = ["inputLayer_pattern_list.0.1.weight", "inputLayer_pattern_list.0.2.weight",
            "inputLayer_pattern_list.0.2.bias", "inputLayer_pattern_list.0.4.weight",
            "inputContact_layer.0.weight", "input_layer_force.1.weight",
            "patternFeatureExtra_layer.0.conv_3_1.0.weight", "patternFeatureExtra_layer.0.conv_3_1.0.bias",
            "patternFeatureExtra_layer.0.conv_5_1.1.weight", "patternFeatureExtra_layer.1.conv_5_2.0.weight",
            "patternFeatureExtra_layer.1.conv_3_2.1.bias", "patternFeatureExtra_layer.1.confusion.weight",
            "patternFeatureExtra_layer.1.confusion.bias", "forceFeatureExtra_layer.0.conv1.weight",
            "forceFeatureExtra_layer.0.conv2.bias", "output_layer.0.weight", "output_layer.2.weight"]
    stat_keys = ["inputLayer_pattern_list.0.2", "patternFeatureExtra_layer.0.conv_5_1.1",
                 "patternFeatureExtra_layer.1.conv_3_2.1", "inputContact_layer.1"]
    cfg = dict(patternFeatureExtraLayerCnt=2)
    seed, B = 211, 4
    m = TactileSR(**cfg)
    sd = load_random(m, seed, **cfg)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-2)
    g = torch.Generator().manual_seed(seed + 1)
    LR = torch.rand(B, 3, 4, 4, generator=g) * 8
    HR_raw = torch.rand(B, 1, 100, 100, generator=g) * 250
    # conditioning yardstick: the same reference step in fp64 (ReLU-mask flips make fp32 gradients of
    # this net differ from exact arithmetic by 1e-4..1e-3 of their max; a faithful fp32 implementation
    # can only be asked to sit as close to fp64 as the reference's own fp32 run does)
    m64 = TactileSR(**cfg).double()
    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    m64.train()
    HR64 = F.interpolate(HR_raw.double() / 10, size=(40, 40), mode="bilinear", align_corners=False)
    l64 = torch.nn.MSELoss()(m64(LR[:, :3].double()), HR64)
    l64.backward()
    named64 = dict(m64.named_parameters())
    for k in keys:
        gk = named64[k].grad
        out[f"grad64/{k}"] = gk.flatten()[:: max(1, gk.numel() // 512)].numpy().copy()
    out["loss64"] = np.float64(l64.item())
    losses = []
    named = dict(m.named_parameters())
    for step in range(2):
        HR = HR_raw.type(torch.float32) / 10
        HR = F.interpolate(HR, size=(40, 40), mode="bilinear", align_corners=False)
        y = m(LR[:, :3])
        loss = torch.nn.MSELoss()(y, HR)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["HR_prepared"] = HR.numpy()
            out["out0"] = y.detach().numpy()
            for k in keys:
                gk = named[k].grad
                out[f"grad/{k}"] = gk.flatten()[:: max(1, gk.numel() // 512)].numpy().copy()
                out[f"gradnorm/{k}"] = np.float64(gk.double().norm().item())
        opt.step()
        losses.append(loss.item())
        if step == 0:
            new_sd = m.state_dict()
            for k in keys:
                w = new_sd[k]
                out[f"w1/{k}"] = w.flatten()[:: max(1, w.numel() // 512)].numpy().copy()
            for s in stat_keys:
                out[f"stat/{s}.running_mean"] = new_sd[s + ".running_mean"].numpy().copy()
                out[f"stat/{s}.running_var"] = new_sd[s + ".running_var"].numpy().copy()
                out[f"stat/{s}.num_batches_tracked"] = new_sd[s + ".num_batches_tracked"].numpy().copy()
    out["seed"] = np.int64(seed)
    out["sha256"] = np.array(sd_hash(sd))
    out["LR"] = LR.numpy()
    out["HR_raw"] = HR_raw.numpy()
    out["losses"] = np.array(losses, np.float64)
    out["keys"] = np.array(keys)
    out["stat_keys"] = np.array(stat_keys)
    print("train losses", losses)
    np.savez_compressed(os.path.join(HERE, "train.npz"), **out)


def gen_tpsf():
    """F6: tPSFNet forward (4 outputs) + MLP gradients of the trainer loss
    (train/tPSFNet_train.py:180-190)."""
    out = {}
    seed, B = 311, 4
    net = tPSFNet(gama=1.4, perception_scale=None, device="cpu")
    shapes = O.tpsf_state_shapes()
    assert list(net.state_dict().keys()) == list(shapes.keys())
    sd = O.random_state_dict(shapes, seed)
    net.load_state_dict(sd, strict=True)
    out["sha256"] = np.array(sd_hash(sd))
    out["PSF_sdf"] = net.PSF_sdf[0, 0, ::7, ::7].numpy()
    out["LR_masking_sdf"] = net.LR_masking_sdf[:, :, ::9, ::9].numpy()
    out["PSF_sdf_sum"] = np.float64(net.PSF_sdf.double().sum().item())
    out["LR_masking_sdf_sum"] = np.float64(net.LR_masking_sdf.double().sum().item())
    g = torch.Generator().manual_seed(seed + 1)
    depth = (torch.rand(B, 100, 100, generator=g) > 0.7).float()
    # second half of the batch: smooth blobs with a genuine plateau
    yy, xx = torch.meshgrid(torch.arange(100.0), torch.arange(100.0), indexing="ij")
    depth[2] = torch.clamp(1.5 - ((yy - 40) ** 2 + (xx - 55) ** 2) ** 0.5 / 20, 0, 1)
    depth[3] = torch.clamp(2.0 - ((yy - 70) ** 2 / 2 + (xx - 30) ** 2) ** 0.5 / 12, 0, 1.25)
    LR_raw = torch.rand(B, 3, 4, 4, generator=g) * 800
    LR = LR_raw.type(torch.float32) / 100
    HR, LRd, psf, ab = net(LR, depth.unsqueeze(1))
    loss = torch.nn.MSELoss()(LR[:, 2:3], LRd)
    loss.backward()
    out["seed"] = np.int64(seed)
    out["depth"] = depth.numpy()
    out["LR_raw"] = LR_raw.numpy()
    out["HR"] = HR.detach().numpy()
    out["LR_deg"] = LRd.detach().numpy()
    out["psf_probe"] = psf.detach()[:, 0, ::7, ::7].numpy()
    out["psf_sum"] = psf.detach().double().sum(dim=(1, 2, 3)).numpy()
    out["alphaBeta"] = ab.detach().numpy()
    out["loss"] = np.float64(loss.item())
    for k, v in net.named_parameters():
        out[f"grad/{k}"] = v.grad.numpy()
    print("tpsf loss", loss.item(), "alphaBeta", ab.detach().view(B, 3)[0])
    np.savez_compressed(os.path.join(HERE, "tpsf.npz"), **out)


def gen_metrics():
    """F7/F8: PSNR (incl. the /40 quirk), SSIM, bilinear tables."""
    out = {}
    g = torch.Generator().manual_seed(411)
    a = torch.rand(3, 1, 40, 40, generator=g) * 25
    b = a + torch.randn(3, 1, 40, 40, generator=g) * 0.3
    out["a"], out["b"] = a.numpy(), b.numpy()
    out["psnr_140"] = np.array([calculationPSNR(a[i], b[i], maxValue=250).item() for i in range(3)])
    out["psnr_40"] = np.array([calculationPSNR(a[i, 0], b[i, 0], maxValue=250).item() for i in range(3)])
    out["ssim_140"] = np.array([calculationSSIM(a[i], b[i]).item() for i in range(3)])
    out["ssim_40"] = np.array([calculationSSIM(a[i, 0], b[i, 0]).item() for i in range(3)])
    x = torch.rand(2, 3, 4, 4, generator=g) * 8
    out["up_in"] = x.numpy()
    out["up_4_40"] = torch.nn.Upsample(scale_factor=10, mode="bilinear", align_corners=False)(x).numpy()
    out["up_4_100"] = torch.nn.Upsample(scale_factor=25, mode="bilinear", align_corners=False)(x).numpy()
    h = torch.rand(2, 1, 100, 100, generator=g) * 250
    out["down_in"] = h.numpy()
    out["down_100_40"] = F.interpolate(h, size=(40, 40), mode="bilinear", align_corners=False).numpy()
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)




def gen_lr():
    """LR sequences of the reference's own LRWarmupScheduler (cpu/lr_scheduler.py, loaded by path because
    cpu/__init__ needs tensorboard) wrapped around StepLR: one value after every iter_update / epoch_update."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_lr", "/root/reference/cpu/lr_scheduler.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {k: lr_sequence(ref.LRWarmupScheduler, dict(c), 6, 500) for k, c in LR_CASES.items()}
    np.savez_compressed(os.path.join(HERE, "lr_schedule.npz"), **out)


def gen_srcnn():
    """TactileSRCNN (model/tactileSR_model.py:101-153; imported at train/tactileSR_train.py:24, instantiated nowhere):
    the seeded construction (state_dict hash: init RNG order) and an eval forward with randomised parameters."""
    from model.tactileSR_model import TactileSRCNN
    out = {}
    torch.manual_seed(42)
    m = TactileSRCNN()
    sd = m.state_dict()
    assert list(sd.keys()) == list(O.tactilesrcnn_state_shapes().keys()), "state_dict key order mismatch"
    out["init_sha"] = np.frombuffer(bytes.fromhex(sd_hash(sd)), dtype=np.uint8)
    rs = O.random_state_dict(O.tactilesrcnn_state_shapes(), 909)
    m.load_state_dict(rs, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(910)
    x = torch.rand(2, 3, 4, 4, generator=g) * 8
    with torch.no_grad():
        y = m(x)
        y64 = copy.deepcopy(m).double()(x.double())
    out.update(seed=np.int64(909), x=x.numpy(), y=y.numpy(), y64=y64.numpy())
    np.savez_compressed(os.path.join(HERE, "srcnn.npz"), **out)


def gen_lr_short():
    """A SHORT schedule of the reference's own LRWarmupScheduler for the -m gpu trainer-loop test (the shipped one needs
    2000 iterations before StepLR shows): 'auto' warm-up over 8 iterations in front of StepLR(1, 0.8), epoch_len = 6,
    3 epochs; one value after every iter_update / epoch_update, like gen_lr."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_lr", "/root/reference/cpu/lr_scheduler.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    np.savez_compressed(os.path.join(HERE, "lr_schedule_short.npz"),
                        short_auto=lr_sequence(ref.LRWarmupScheduler, dict(LR_SHORT), 3, 6))


def gen_config():
    """Keys and values of the reference's three config dicts (config/default.py:8-96).  The module cannot be imported
    (it shells out to nvidia-smi at import, :101-104), so its dict literals are evaluated from the file's text up to
    that point; '/code' path prefixes are stored as '<root>'."""
    import json
    src = open("/root/reference/config/default.py").read().split("# TODO: change to another file")[0]
    src = src.replace("from utility.tools import select_gpu_with_least_used_memory", "")
    ns = {}
    exec(compile(src, "reference-config", "exec"), ns)
    out = {}
    for name in ("tPSFNet_config", "tactileSR_config", "tactileSeqs_config"):
        out[name] = {k: ("<root>" + v[len("/code"):] if isinstance(v, str) and v.startswith("/code") else v)
                     for k, v in ns[name].items()}
    with open(os.path.join(HERE, "config_default.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def gen_lr_state():
    """State dict of the reference's OWN warm-up scheduler 700 ticks into the shipped schedule (inside the warm-up),
    together with the optimizer state at that point: what a reference-written checkpoint holds under
    'lr_scheduler' / 'optimizer' (cpu/trainer.py:401-411).  Saved with torch.save (tensors, lists, dicts, numbers
    only: loads with weights_only=True)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_lr", "/root/reference/cpu/lr_scheduler.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cfg = dict(LR_CASES["auto_shipped"])
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=cfg.pop('lr'))
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.pop('step'), gamma=0.8)
    w = ref.LRWarmupScheduler(sch, epoch_len=500, **cfg)
    tick = 0
    for _ in range(6):
        for _ in range(500):
            tick += 1
            if tick < 701:
                p.grad = torch.ones(1)
                opt.step()
                w.iter_update()
        tick += 1
        if tick < 701:
            w.epoch_update()
    torch.save({"lr_scheduler": w.state_dict(), "optimizer": opt.state_dict(), "ticks": 700},
               os.path.join(HERE, "ref_lr_state.pth"))


LR_CASES = {
    "auto_shipped": dict(lr=1e-3, step=2, by_epoch=True, warmup_t=2000, warmup_by_epoch=False, warmup_mode='auto',
                         warmup_init_lr=1e-5, warmup_factor=1e-4),      # config/default.py:56-60 as forwarded
    "fix": dict(lr=1e-3, step=2, by_epoch=True, warmup_t=700, warmup_by_epoch=False, warmup_mode='fix',
                warmup_init_lr=1e-5, warmup_factor=1e-4),
    "factor": dict(lr=1e-3, step=1, by_epoch=True, warmup_t=1300, warmup_by_epoch=False, warmup_mode='factor',
                   warmup_init_lr=1e-5, warmup_factor=1e-2),
    "by_epoch": dict(lr=1e-4, step=1, by_epoch=True, warmup_t=3, warmup_by_epoch=True, warmup_mode='auto',
                     warmup_init_lr=1e-5, warmup_factor=1e-1),
    "none": dict(lr=1e-4, step=1, by_epoch=True, warmup_t=0, warmup_by_epoch=False, warmup_mode='fix',
                 warmup_init_lr=0.0, warmup_factor=0.0),
}


LR_SHORT = dict(lr=1e-4, step=1, by_epoch=True, warmup_t=8, warmup_by_epoch=False, warmup_mode='auto',
                warmup_init_lr=1e-5, warmup_factor=1e-2)


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
    return np.array(lrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["init", "eval", "train", "tpsf", "metrics", "lr", "lr_short", "lr_state", "config", "srcnn"]
    for w in which:
        globals()["gen_" + w]()
        print("wrote", w)
