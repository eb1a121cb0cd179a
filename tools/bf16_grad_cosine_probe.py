#!/usr/bin/env python3
"""Worst per-parameter gradient cosine of the bf16-storage train step against the bf16-emulating oracle, two-stem model, for a
few seeds / batch sizes:   python tools/bf16_grad_cosine_probe.py {all|no64}     (no64: the 64-channel 3x3 / 5x5 layers on the
32x32x16 kernels instead of conv_b16k).  Background of the B = 11 case of test_train_step_bf16_storage_vs_bf16_emulating_oracle."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import tactilesr_oracle as O
from tactilesr_amd.model import tactileSR_model as T
from tactilesr_amd.model import _train
mode = sys.argv[1]
if mode == "no64":
    lib = _train._lib.load(); real = lib.tsr_conv2d_ex_dgrad_b16k
    lib.tsr_conv2d_ex_dgrad_b16k = lambda n, c, k: 0 if (n == 64 and k > 1) else real(n, c, k)
def emu(sd, LR, HR, **kw):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if O.is_trainable(k)}
    full = dict(sd); full.update(leaves)
    out = O.tactilesr_forward(full, LR, training=True, new_stats={}, emulate="bf16", **kw)
    loss = F.mse_loss(out, HR)
    gl = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    return float(loss), {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(leaves, gl)}
for cfg, B, seed in [(dict(seqsCnt=2, patternFeatureExtraLayerCnt=1), 3, 977), (dict(seqsCnt=2, patternFeatureExtraLayerCnt=1), 11, 977),
                     (dict(seqsCnt=2, patternFeatureExtraLayerCnt=1), 11, 978), (dict(seqsCnt=2, patternFeatureExtraLayerCnt=1), 11, 979)]:
    sd = O.random_state_dict(O.tactilesr_state_shapes(**cfg), seed)
    g = torch.Generator().manual_seed(seed + 1)
    LR = torch.rand(B, 6, 4, 4, generator=g) * 8
    HR = torch.rand(B, 1, 40, 40, generator=g) * 25
    l_e, g_e = emu(sd, LR, HR)
    m = T.TactileSR(**cfg); m.train_impl = "bf16"; m.load_state_dict(sd, strict=True); m = m.cuda().train()
    loss = F.mse_loss(m(LR.cuda()), HR.cuda()); loss.backward()
    gm = float(max(v.abs().max() for v in g_e.values()))
    cs = []
    for k, p in m.named_parameters():
        ref = g_e[k].double().flatten()
        if float(ref.abs().max()) < 1e-6 * gm: continue
        got = p.grad.detach().cpu().double().flatten()
        cs.append((float(got @ ref / (got.norm() * ref.norm()).clamp_min(1e-30)), k))
    cs.sort()
    print(mode, seed, B, "loss", abs(loss.item() - l_e) / abs(l_e), "worst", [(round(c, 5), k) for c, k in cs[:3]])
