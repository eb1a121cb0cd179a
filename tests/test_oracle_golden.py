"""Pin the CPU oracle (oracle/tactilesr_oracle.py) against the golden fixtures that
tests/golden/make_golden.py produced by running the reference itself."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import tactilesr_oracle as O

torch.set_num_threads(8)


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def probe(t):
    cs = max(1, t.shape[1] // 4)
    return t[:, ::cs, ::3, ::3].contiguous().numpy()


CFGS = {"t1": dict(), "t7": dict(seqsCnt=7), "sf25t8": dict(scale_factor=25, seqsCnt=8),
        "t1_l2": dict(patternFeatureExtraLayerCnt=2)}


@pytest.mark.parametrize("tag", ["t1", "t7", "t1_l2", "sf25t8"])
def test_eval_forward_matches_reference(golden, tag):
    g = golden("eval")
    cfg = CFGS[tag]
    sd = O.random_state_dict(O.tactilesr_state_shapes(**cfg), int(g[f"{tag}/seed"]))
    assert sd_hash(sd) == str(g[f"{tag}/sha256"]), "seeded parameters drifted (torch RNG changed?)"
    stages = {}
    with torch.no_grad():
        y = O.tactilesr_forward(sd, torch.from_numpy(g[f"{tag}/LR"]), cfg.get("scale_factor", 10),
                                stages=stages)
    if tag == "sf25t8":
        assert relerr(probe(y), g[f"{tag}/out/probe"]) < 1e-6
        assert relerr(y[0, 0, ::2, ::2].numpy(), g[f"{tag}/out_full0"]) < 1e-6
    else:
        assert relerr(y.numpy(), g[f"{tag}/out"]) < 1e-6
    for name, t in stages.items():
        assert relerr(probe(t), g[f"{tag}/stage/{name}/probe"]) < 1e-6, name
        assert abs(float(t.double().sum()) - float(g[f"{tag}/stage/{name}/sum"])) \
            <= 1e-5 * float(g[f"{tag}/stage/{name}/abssum"]), name


INIT_CFGS = {"init_t1": dict(), "init_sf25t8": dict(scale_factor=25, seqsCnt=8)}


def init_fixture_state(g, tag):
    """Parameters of the `eval_init` fixtures: the drop-in's constructor under seed 42 (bit-identical to the reference's
    `_init_network`, checked by sha256 against the reference's own state_dict) + the running statistics the reference's
    two train-mode passes left behind (stored in the fixture)."""
    import tactilesr_amd
    torch.manual_seed(42)
    sd = {k: v.detach().clone() for k, v in tactilesr_amd.TactileSR(**INIT_CFGS[tag]).state_dict().items()}
    assert sd_hash(sd) == str(g[f"{tag}/sha256_init"]), "seeded construction differs from the reference's"
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            sd[k] = torch.from_numpy(g[f"{tag}/stat/{k}"])
    return sd


@pytest.mark.parametrize("tag", ["init_t1", "init_sf25t8"])
def test_eval_forward_reference_init_fixture(golden, tag):
    """The reference's own seed-42 parameters with moved BatchNorm statistics (configs[4] shape and the shipped shape)."""
    g = golden("eval_init")
    sd = init_fixture_state(g, tag)
    stages = {}
    with torch.no_grad():
        y = O.tactilesr_forward(sd, torch.from_numpy(g[f"{tag}/LR"]), INIT_CFGS[tag].get("scale_factor", 10),
                                stages=stages)
    assert relerr(y.numpy(), g[f"{tag}/out"]) < 1e-6
    for name, t in stages.items():
        assert relerr(probe(t), g[f"{tag}/stage/{name}/probe"]) < 1e-6, name


def test_bf16_emulation_hooks(golden):
    """`emulate="bf16"` (the restatement of the build's bf16 activation-storage arithmetic): every recorded stage holds
    bf16-representable values, the result stays within the stated 3e-2 of the reference's fp32 output, feeding the
    oracle's own stages back as `teacher` changes nothing, and a perturbed teacher tensor changes only what follows."""
    g = golden("eval")
    cfg = CFGS["t1_l2"]
    sd = O.random_state_dict(O.tactilesr_state_shapes(**cfg), int(g["t1_l2/seed"]))
    LR = torch.from_numpy(g["t1_l2/LR"])
    st = {}
    with torch.no_grad():
        y = O.tactilesr_forward(sd, LR, stages=st, emulate="bf16")
        y32 = O.tactilesr_forward(sd, LR)
    assert torch.equal(y32, torch.from_numpy(g["t1_l2/out"])) or relerr(y32.numpy(), g["t1_l2/out"]) < 1e-6
    for name, t in st.items():
        assert torch.equal(t, t.to(torch.bfloat16).float()), name
    assert 1e-4 < relerr(y.numpy(), y32.numpy()) < 3e-2
    st2 = {}
    with torch.no_grad():
        y2 = O.tactilesr_forward(sd, LR, stages=st2, emulate="bf16", teacher=st)
    assert torch.equal(y, y2) and all(torch.equal(st[k], st2[k]) for k in st)
    bad = dict(st)
    bad["msrb0"] = st["msrb0"] * 1.5
    st3 = {}
    with torch.no_grad():
        O.tactilesr_forward(sd, LR, stages=st3, emulate="bf16", teacher=bad)
    assert torch.equal(st3["msrb0"], st["msrb0"]) and torch.equal(st3["fuse"], st["fuse"])
    assert not torch.equal(st3["msrb1"], st["msrb1"])


def test_train_step_matches_reference(golden):
    g = golden("train")
    cfg = dict(patternFeatureExtraLayerCnt=2)
    p = O.random_state_dict(O.tactilesr_state_shapes(**cfg), int(g["seed"]))
    assert sd_hash(p) == str(g["sha256"])
    LR, HR_raw = torch.from_numpy(g["LR"]), torch.from_numpy(g["HR_raw"])
    assert relerr(O.prepare_target(HR_raw).numpy(), g["HR_prepared"]) < 1e-6
    state = {}
    loss0, grads = O.train_one_iter(p, state, 1, LR, HR_raw, lr=1e-3, weight_decay=1e-2)
    assert abs(loss0 - g["losses"][0]) <= 1e-5 * abs(g["losses"][0])
    for k in [str(k) for k in g["keys"]]:
        gk = grads[k]
        ref = g[f"grad/{k}"]
        got = gk.flatten()[:: max(1, gk.numel() // 512)].numpy()
        scale = float(g[f"gradnorm/{k}"]) / np.sqrt(gk.numel()) + 1e-30
        assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), scale), k
        w = p[k]
        assert relerr(w.flatten()[:: max(1, w.numel() // 512)].numpy(), g[f"w1/{k}"]) < 1e-5, k
    for s in [str(s) for s in g["stat_keys"]]:
        assert relerr(p[s + ".running_mean"].numpy(), g[f"stat/{s}.running_mean"]) < 1e-5, s
        assert relerr(p[s + ".running_var"].numpy(), g[f"stat/{s}.running_var"]) < 1e-5, s
        assert int(p[s + ".num_batches_tracked"]) == int(g[f"stat/{s}.num_batches_tracked"])
    loss1, _ = O.train_one_iter(p, state, 2, LR, HR_raw, lr=1e-3, weight_decay=1e-2)
    assert abs(loss1 - g["losses"][1]) <= 1e-4 * abs(g["losses"][1])


def test_metrics_and_bilinear(golden):
    g = golden("metrics")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    for i in range(3):
        assert abs(float(O.calculation_psnr(a[i], b[i], 250)) - g["psnr_140"][i]) < 1e-4
        assert abs(float(O.calculation_psnr(a[i, 0], b[i, 0], 250)) - g["psnr_40"][i]) < 1e-4
        # the (1,40,40) call divides by 40 instead of 1600: 10*log10(40) dB lower
        assert abs((g["psnr_40"][i] - g["psnr_140"][i]) - 10 * np.log10(40)) < 1e-3
        assert abs(float(O.calculation_ssim(a[i], b[i])) - g["ssim_140"][i]) < 1e-6
        assert abs(float(O.calculation_ssim(a[i, 0], b[i, 0])) - g["ssim_40"][i]) < 1e-6
    x = torch.from_numpy(g["up_in"])
    assert np.array_equal(O.bilinear_resize(x, (40, 40)).numpy(), g["up_4_40"])
    assert relerr(O.bilinear_resize_table(x, (40, 40)).numpy(), g["up_4_40"]) < 5e-7
    assert relerr(O.bilinear_resize_table(x, (100, 100)).numpy(), g["up_4_100"]) < 5e-7
    h = torch.from_numpy(g["down_in"])
    assert relerr(O.bilinear_resize_table(h, (40, 40)).numpy(), g["down_100_40"]) < 5e-7


def test_tpsf_matches_reference(golden):
    g = golden("tpsf")
    p = O.random_state_dict(O.tpsf_state_shapes(), int(g["seed"]))
    assert sd_hash(p) == str(g["sha256"])
    geom = O.tpsf_geometry()
    assert relerr(geom[0][0, 0, ::7, ::7].numpy(), g["PSF_sdf"]) < 1e-6
    assert relerr(geom[1][:, :, ::9, ::9].numpy(), g["LR_masking_sdf"]) < 1e-6
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    LR_raw, depth = torch.from_numpy(g["LR_raw"]), torch.from_numpy(g["depth"])
    HR, LRd, psf, ab = O.tpsf_forward(leaves, LR_raw / 100, depth.unsqueeze(1), geom)
    assert relerr(HR.detach().numpy(), g["HR"]) < 1e-6
    assert relerr(LRd.detach().numpy(), g["LR_deg"]) < 1e-6
    assert relerr(ab.detach().numpy(), g["alphaBeta"]) < 1e-6
    assert relerr(psf.detach()[:, 0, ::7, ::7].numpy(), g["psf_probe"]) < 1e-6
    loss = O.tpsf_train_cal_loss(leaves, LR_raw, depth, 100.0, geom)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    for k, v in leaves.items():
        ref = g[f"grad/{k}"]
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-12, k


def test_conv_f64_truth_small():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 5, 7, 6, generator=g)
    w = torch.randn(4, 5, 3, 3, generator=g)
    b = torch.randn(4, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).numpy()
    assert np.abs(O.conv2d_f64(x.numpy(), w.numpy(), b.numpy(), 1) - ref).max() < 1e-12
