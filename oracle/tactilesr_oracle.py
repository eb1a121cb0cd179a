"""CPU oracle for the tactileSR hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain ``torch.nn.functional`` ops on CPU fp32
tensors, parameters held in a flat ``dict`` keyed by the reference's
``state_dict`` names) of the algorithm the reference runs through its
``nn.Module`` classes.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package ``tactilesr_amd`` never imports anything from
``oracle/`` and fails loudly when its HIP library is missing.

Parity pin: the reference holds no tests / golden vectors, so
the oracle is pinned against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference/model/*.py``) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every function below against them.

Reference citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default (model/tactileSR_model.py:38)
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default


# --------------------------------------------------------------------------
# state_dict layout (model/tactileSR_model.py:29-63,167-191,219-220)
# --------------------------------------------------------------------------
def tactilesr_state_shapes(scale_factor=10, seqsCnt=1, axisCnt=3,
                           patternFeatureExtraLayerCnt=6,
                           forceFeatureExtraLayerCnt=1) -> "Dict[str, Tuple[int, ...]]":
    """Key -> shape in the exact ``state_dict()`` order of the reference class
    (registration order: patternFeatureExtra_layer, forceFeatureExtra_layer,
    inputLayer_pattern_list, inputContact_layer, output_layer,
    input_layer_force; model/tactileSR_model.py:29-63)."""
    d: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix, c):
        d[prefix + ".weight"] = (c,)
        d[prefix + ".bias"] = (c,)
        d[prefix + ".running_mean"] = (c,)
        d[prefix + ".running_var"] = (c,)
        d[prefix + ".num_batches_tracked"] = ()

    for i in range(patternFeatureExtraLayerCnt):
        p = f"patternFeatureExtra_layer.{i}"
        for name, c, k in (("conv_3_1", 64, 3), ("conv_5_1", 64, 5),
                           ("conv_3_2", 128, 3), ("conv_5_2", 128, 5)):
            d[f"{p}.{name}.0.weight"] = (c, c, k, k)
            d[f"{p}.{name}.0.bias"] = (c,)
            bn(f"{p}.{name}.1", c)
        d[f"{p}.confusion.weight"] = (64, 256, 1, 1)
        d[f"{p}.confusion.bias"] = (64,)
    for i in range(forceFeatureExtraLayerCnt):
        p = f"forceFeatureExtra_layer.{i}"
        for name in ("conv1", "conv2"):
            d[f"{p}.{name}.weight"] = (64, 64, 3, 3)
            d[f"{p}.{name}.bias"] = (64,)
    for t in range(seqsCnt):
        p = f"inputLayer_pattern_list.{t}"
        d[f"{p}.1.weight"] = (64, axisCnt, 3, 3)
        bn(f"{p}.2", 64)
        d[f"{p}.4.weight"] = (64, 64, 3, 3)
        bn(f"{p}.5", 64)
    d["inputContact_layer.0.weight"] = (64, 64 * seqsCnt, 3, 3)
    bn("inputContact_layer.1", 64)
    d["output_layer.0.weight"] = (128, 128, 3, 3)
    d["output_layer.2.weight"] = (1, 128, 3, 3)
    d["input_layer_force.1.weight"] = (64, axisCnt, 3, 3)
    return d


def random_state_dict(shapes: "Dict[str, Tuple[int, ...]]", seed: int) -> Params:
    """Deterministic, well-conditioned random parameters used by the golden
    fixtures (the reference ships no checkpoints: .MISSING_LARGE_BLOBS).  Conv
    weights ~ N(0, 2/fan_out) (the law of model/tactileSR_model.py:95); BN
    affine / running stats are randomised so that every folded constant is
    exercised (fresh BN stats 0/1 would hide mean/var handling)."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(3, dtype=torch.int64)
        elif k.endswith("running_var"):
            out[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("running_mean"):
            out[k] = torch.randn(shp, generator=g) * 0.2
        elif len(shp) == 4:
            fan_out = shp[0] * shp[2] * shp[3]
            out[k] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_out)
        elif len(shp) == 2:
            out[k] = torch.randn(shp, generator=g) * 0.03
        elif k.endswith(".1.weight") or k.endswith(".2.weight") or k.endswith(".5.weight"):
            # BN gamma (1-D "weight")
            out[k] = torch.rand(shp, generator=g) + 0.5
        else:
            out[k] = torch.randn(shp, generator=g) * 0.1   # biases / BN beta
    return out


# --------------------------------------------------------------------------
# bilinear resize (nn.Upsample / F.interpolate, align_corners=False)
# model/tactileSR_model.py:35,60,83 ; train/tactileSR_train.py:45
# --------------------------------------------------------------------------
def bilinear_table(n_in: int, n_out: int) -> "Tuple[np.ndarray, np.ndarray, np.ndarray]":
    """(i0, i1, lam) per output index for ATen's upsample_bilinear2d with
    align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0, scale=n_in/n_out,
    i0=floor(src), i1=min(i0+1,n_in-1), lam=src-i0.  All in fp32 like ATen."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0)).astype(np.float32)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def bilinear_resize(x: torch.Tensor, out_hw: "Tuple[int, int]") -> torch.Tensor:
    return F.interpolate(x, size=out_hw, mode="bilinear", align_corners=False)


def bilinear_resize_table(x: torch.Tensor, out_hw: "Tuple[int, int]") -> torch.Tensor:
    """Table restatement of ``bilinear_resize`` (what the HIP kernels implement):
    out = (1-ly)*((1-lx)*a + lx*b) + ly*((1-lx)*c + lx*d)."""
    H, W = x.shape[-2:]
    y0, y1, ly = bilinear_table(H, out_hw[0])
    x0, x1, lx = bilinear_table(W, out_hw[1])
    ly = torch.from_numpy(ly).view(-1, 1)
    lx = torch.from_numpy(lx).view(1, -1)
    a = x[..., y0, :][..., :, x0]
    b = x[..., y0, :][..., :, x1]
    c = x[..., y1, :][..., :, x0]
    d = x[..., y1, :][..., :, x1]
    return (1 - ly) * ((1 - lx) * a + lx * b) + ly * ((1 - lx) * c + lx * d)


# --------------------------------------------------------------------------
# TactileSR forward  (model/tactileSR_model.py:67-84)
# --------------------------------------------------------------------------
def _bn(p: Params, prefix: str, x: torch.Tensor, training: bool,
        new_stats: Optional[Params]) -> torch.Tensor:
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1): train = batch statistics (biased var to
    normalise, unbiased var into running_var), eval = running statistics."""
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"],
                         True, BN_MOMENTUM, BN_EPS)
        if new_stats is not None:
            new_stats[prefix + ".running_mean"] = rm
            new_stats[prefix + ".running_var"] = rv
            new_stats[prefix + ".num_batches_tracked"] = p[prefix + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"],
                        False, BN_MOMENTUM, BN_EPS)


class ReluTap:
    """Optional instrumentation of every ReLU of the network (tests only).

    ``pre`` receives the pre-activation of each ReLU under its name; ``masks`` (name -> bool tensor)
    REPLACES ``relu(x)`` by ``x * mask``: the forward value changes only where the given mask disagrees
    with ``x > 0`` (|x| ~ one ulp of zero there), and the backward pass routes gradients through
    exactly the given mask.  The GPU gradient tests use this to evaluate the fp64 gradient *on the
    activation pattern the device actually took* (two faithful fp32 forwards disagree on the sign of
    pre-activations that are zero to rounding; the gradient is discontinuous there).

    Names: ``<bn prefix>`` for conv+BN+ReLU (e.g. ``patternFeatureExtra_layer.0.conv_3_1.1``),
    ``<block prefix>.out`` for the MSRB / ResBlock output ReLU, ``<resblock prefix>.conv1`` for the
    inner ResBlock ReLU, ``force_in``, ``head0`` and ``out``."""

    def __init__(self, masks: Optional[Dict[str, torch.Tensor]] = None, record: bool = False):
        self.masks = masks
        self.pre: Optional[Dict[str, torch.Tensor]] = {} if record else None

    def __call__(self, name: str, x: torch.Tensor) -> torch.Tensor:
        if self.pre is not None:
            self.pre[name] = x.detach()
        if self.masks is not None:
            return x * self.masks[name].to(x.dtype)
        return F.relu(x)


def _relu(tap: Optional[ReluTap], name: str, x: torch.Tensor) -> torch.Tensor:
    return F.relu(x) if tap is None else tap(name, x)


# ---- reduced-precision emulation (BASELINE's "bf16" configurations) ----------------------------------------------
# The reference has no bf16 numerics (cpu/trainer.py:96,203,346-362 only carries an unused fp16 autocast switch), so
# the bf16 paths of the build are checked against THIS restatement of their arithmetic: every tensor the device
# stores in HBM as bf16 is rounded to bf16 (round-to-nearest-even) at the same point, conv weights are rounded to
# bf16 once, products are exact and accumulation is wide (float64 here, fp32 inside the MFMA on the device), the
# epilogue arithmetic (BatchNorm fold, bias, residual, ReLU) is fp32 on fp32 parameters.  `emulate=None` is the
# reference's plain fp32 path and leaves every function below exactly as it was.
def _q(t: torch.Tensor, emulate) -> torch.Tensor:
    """Round to the storage type of the emulated path (identity for emulate=None)."""
    if emulate is None:
        return t
    assert emulate == "bf16", emulate
    return t.to(torch.bfloat16).to(t.dtype)


def _conv_q(x, w, bias, pad, emulate):
    """conv2d with the operand handling of the emulated path: weights rounded to the storage type, exact products,
    wide accumulation; the input must already hold storage-type values."""
    if emulate is None:
        return F.conv2d(x, w, bias, padding=pad)
    y = F.conv2d(x.double(), _q(w, emulate).double(), None, padding=pad).to(x.dtype)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def _bn_train_stored(p: Params, prefix: str, z: torch.Tensor, bias, new_stats, emulate, stats_from_stored=False):
    """Train-mode BatchNorm as the build's bf16-storage train path evaluates it: `z` is the bias-free conv output in
    fp32; the batch statistics come from the fp32 accumulator (`stats_from_stored`: from the stored, rounded tensor --
    the stem, whose statistics are a separate pass over what it wrote), the normalisation is applied to the STORED
    (rounded) z as fma(zq, scale, shift) with scale = gamma * invstd, shift = beta - mean * scale; the conv bias only
    shifts running_mean."""
    zq = _q(z, emulate)
    src = zq if stats_from_stored else z
    mean = src.mean(dim=(0, 2, 3))
    var = src.var(dim=(0, 2, 3), unbiased=False)
    scale = p[prefix + ".weight"] / torch.sqrt(var + BN_EPS)
    shift = p[prefix + ".bias"] - mean * scale
    if new_stats is not None:
        n = z.numel() // z.shape[1]
        mz = mean.detach() + (bias.detach() if bias is not None else 0.0)
        new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * p[prefix + ".running_mean"] + BN_MOMENTUM * mz
        new_stats[prefix + ".running_var"] = ((1 - BN_MOMENTUM) * p[prefix + ".running_var"]
                                              + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
        new_stats[prefix + ".num_batches_tracked"] = p[prefix + ".num_batches_tracked"] + 1
    return zq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def _conv_bn_relu(p, prefix_conv, prefix_bn, x, pad, training, new_stats, tap=None, emulate=None,
                  fp32_conv=False, stats_from_stored=False):
    """conv (+bias) -> BatchNorm -> ReLU.  `fp32_conv`: the stem's 3 -> 64 conv keeps fp32 operands in the emulated
    paths too (it is fused with the bilinear upsample of the fp32 taxels)."""
    w, b = p[prefix_conv + ".weight"], p.get(prefix_conv + ".bias")
    if emulate is not None and training:
        z = F.conv2d(x, w, None, padding=pad) if fp32_conv else _conv_q(x, w, None, pad, emulate)
        return _q(_relu(tap, prefix_bn, _bn_train_stored(p, prefix_bn, z, b, new_stats, emulate, stats_from_stored)), emulate)
    y = F.conv2d(x, w, b, padding=pad) if fp32_conv else _conv_q(x, w, b, pad, emulate)
    return _q(_relu(tap, prefix_bn, _bn(p, prefix_bn, y, training, new_stats)), emulate)


def msrb_forward(p: Params, prefix: str, x: torch.Tensor, training=False,
                 new_stats: Optional[Params] = None, tap: Optional[ReluTap] = None, emulate=None) -> torch.Tensor:
    """MSRB.forward, model/tactileSR_model.py:196-206.  `emulate="bf16"`: the inference path with bf16 activation
    storage applies each 128-channel half of the 1x1 `confusion` inside the stage-2 launch that produced its input
    (the 3x3 half first, its partial sum + bias + x stored as bf16, then the 5x5 half + ReLU), so the rounding points
    are: cat1, the two stage-2 tiles, the partial sum, the block output."""
    o31 = _conv_bn_relu(p, f"{prefix}.conv_3_1.0", f"{prefix}.conv_3_1.1", x, 1, training, new_stats, tap, emulate)
    o51 = _conv_bn_relu(p, f"{prefix}.conv_5_1.0", f"{prefix}.conv_5_1.1", x, 2, training, new_stats, tap, emulate)
    in2 = torch.cat([o31, o51], 1)
    o32 = _conv_bn_relu(p, f"{prefix}.conv_3_2.0", f"{prefix}.conv_3_2.1", in2, 1, training, new_stats, tap, emulate)
    o52 = _conv_bn_relu(p, f"{prefix}.conv_5_2.0", f"{prefix}.conv_5_2.1", in2, 2, training, new_stats, tap, emulate)
    wc, bc = p[f"{prefix}.confusion.weight"], p[f"{prefix}.confusion.bias"]
    if emulate is not None and not training:
        part = _q(_conv_q(o32, wc[:, :128], bc, 0, emulate) + x, emulate)
        return _q(_relu(tap, f"{prefix}.out", _conv_q(o52, wc[:, 128:], None, 0, emulate) + part), emulate)
    in3 = torch.cat([o32, o52], 1)
    out = _conv_q(in3, wc, bc, 0, emulate)
    return _q(_relu(tap, f"{prefix}.out", out + x), emulate)


def resblock_forward(p: Params, prefix: str, x: torch.Tensor, tap: Optional[ReluTap] = None,
                     emulate=None) -> torch.Tensor:
    """ResBlock.forward, model/tactileSR_model.py:222-225."""
    y = _q(_relu(tap, f"{prefix}.conv1", _conv_q(x, p[f"{prefix}.conv1.weight"], p[f"{prefix}.conv1.bias"], 1, emulate)),
           emulate)
    y = _conv_q(y, p[f"{prefix}.conv2.weight"], p[f"{prefix}.conv2.bias"], 1, emulate)
    return _q(_relu(tap, f"{prefix}.out", x + y), emulate)


def _count(p: Params, stem: str) -> int:
    n = 0
    while any(k.startswith(f"{stem}.{n}.") for k in p):
        n += 1
    return n


def tactilesr_forward(p: Params, x: torch.Tensor, scale_factor=10, axisCnt=3,
                      training=False, new_stats: Optional[Params] = None,
                      stages: Optional[Dict[str, torch.Tensor]] = None,
                      tap: Optional[ReluTap] = None, emulate=None,
                      teacher: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """TactileSR.forward, model/tactileSR_model.py:67-84.  ``stages`` (optional)
    receives named intermediate activations for per-stage parity probes; ``tap``
    (optional, tests only) records / overrides the ReLU activation pattern.
    ``emulate="bf16"`` restates the arithmetic of the build's bf16 activation-storage path (see ``_q``).
    ``teacher`` (optional, tests only; name -> tensor): after stage ``name`` has been computed (and recorded in
    ``stages``) the given tensor -- the device's output of that stage -- is what the following stages consume, so
    that every recorded stage is the oracle's answer to the DEVICE's inputs (a rounding-boundary flip in one stage
    then cannot cascade into the comparison of the next)."""
    seqsCnt = _count(p, "inputLayer_pattern_list")
    n_msrb = _count(p, "patternFeatureExtra_layer")
    n_res = _count(p, "forceFeatureExtra_layer")
    assert x.shape[1] == seqsCnt * axisCnt, "input channel should be same with seqsCnt x axisCnt!"
    size = (x.shape[2] * scale_factor, x.shape[3] * scale_factor)

    def rec(name, t):
        if stages is not None:
            stages[name] = t
        if teacher is not None and name in teacher:
            return teacher[name].to(t.dtype)
        return t

    feats = []
    for t in range(seqsCnt):
        pre = f"inputLayer_pattern_list.{t}"
        u = bilinear_resize(x[:, axisCnt * t:axisCnt * (t + 1)], size)
        # (the stem kernel fuses the upsample with its fp32 3->64 conv: fp32 operands, only its OUTPUT is stored)
        h = _conv_bn_relu(p, f"{pre}.1", f"{pre}.2", u, 1, training, new_stats, tap, emulate, fp32_conv=True,
                          stats_from_stored=True)
        h = _conv_bn_relu(p, f"{pre}.4", f"{pre}.5", h, 1, training, new_stats, tap, emulate)
        feats.append(rec(f"stem{t}", h))
    h = torch.cat(feats, 1) if seqsCnt > 1 else feats[0]
    h = rec("fuse", _conv_bn_relu(p, "inputContact_layer.0", "inputContact_layer.1", h, 1,
                                  training, new_stats, tap, emulate))
    for i in range(n_msrb):
        h = rec(f"msrb{i}", msrb_forward(p, f"patternFeatureExtra_layer.{i}", h, training, new_stats, tap, emulate))
    pattern = h

    u = bilinear_resize(x[:, :axisCnt], size)
    f = rec("force_in", _q(_relu(tap, "force_in", F.conv2d(u, p["input_layer_force.1.weight"], padding=1)), emulate))
    for i in range(n_res):
        f = resblock_forward(p, f"forceFeatureExtra_layer.{i}", f, tap, emulate)
    f = rec("force", f)

    out = torch.cat((f, pattern), 1)           # force first (model/tactileSR_model.py:81)
    out = rec("head0", _q(_relu(tap, "head0", _conv_q(out, p["output_layer.0.weight"], None, 1, emulate)), emulate))
    # (the head kernel reads the stored activations but keeps its 128->1 weights and its output in fp32)
    out = _relu(tap, "out", F.conv2d(out, p["output_layer.2.weight"], padding=1))
    # final F.interpolate to the same size is an exact identity (model/tactileSR_model.py:83)
    out = F.interpolate(out, size=(4 * scale_factor, 4 * scale_factor), mode="bilinear",
                        align_corners=False)
    return out


# --------------------------------------------------------------------------
# TactileSRCNN (model/tactileSR_model.py:101-153): imported by both trainers, instantiated by none
# --------------------------------------------------------------------------
def tactilesrcnn_state_shapes() -> "Dict[str, Tuple[int, ...]]":
    """Key -> shape in the reference class's ``state_dict()`` order (registration order msrb_layer, input_zyx,
    upSample (no state), output; model/tactileSR_model.py:105-129)."""
    d = {k.replace("patternFeatureExtra_layer", "msrb_layer"): v
         for k, v in tactilesr_state_shapes(patternFeatureExtraLayerCnt=6, forceFeatureExtraLayerCnt=0).items()
         if k.startswith("patternFeatureExtra_layer")}
    for i, cin in ((0, 3), (3, 64), (6, 64)):
        d[f"input_zyx.{i}.weight"] = (64, cin, 3, 3)
        for s, shp in (("weight", (64,)), ("bias", (64,)), ("running_mean", (64,)), ("running_var", (64,)),
                       ("num_batches_tracked", ())):
            d[f"input_zyx.{i + 1}.{s}"] = shp
    d["output.0.weight"] = (1, 64, 3, 3)
    return d


def tactilesrcnn_forward(p: Params, x: torch.Tensor) -> torch.Tensor:
    """TactileSRCNN.forward in eval mode, model/tactileSR_model.py:146-151: bilinear x10 -> three conv3x3+BN+ReLU ->
    six MSRBs -> conv 64->1 + ReLU."""
    h = bilinear_resize(x, (x.shape[2] * 10, x.shape[3] * 10))
    for i in (0, 3, 6):
        h = _conv_bn_relu(p, f"input_zyx.{i}", f"input_zyx.{i + 1}", h, 1, False, None)
    for i in range(_count(p, "msrb_layer")):
        h = msrb_forward(p, f"msrb_layer.{i}", h)
    return F.relu(F.conv2d(h, p["output.0.weight"], padding=1))


# --------------------------------------------------------------------------
# trainer step semantics (train/tactileSR_train.py:41-51 ; cpu/trainer.py:346-362)
# --------------------------------------------------------------------------
def prepare_target(HR_raw: torch.Tensor, HR_scale_num=10.0, scale_factor=10) -> torch.Tensor:
    """HR.float()/HR_scale_num then bilinear to (4*sf, 4*sf); train/tactileSR_train.py:44-45."""
    HR = HR_raw.type(torch.float32) / HR_scale_num
    return bilinear_resize(HR, (4 * scale_factor, 4 * scale_factor))


def train_cal_loss(p: Params, LR: torch.Tensor, HR_raw: torch.Tensor, seqsCnt=1, axisCnt=3,
                   HR_scale_num=10.0, scale_factor=10, training=True,
                   new_stats: Optional[Params] = None, emulate=None) -> torch.Tensor:
    """Trainer_tactileSR.train_cal_loss, train/tactileSR_train.py:41-51."""
    HR = prepare_target(HR_raw, HR_scale_num, scale_factor)
    x = LR.type(torch.float32)[:, :seqsCnt * axisCnt]
    out = tactilesr_forward(p, x, scale_factor, axisCnt, training, new_stats, emulate=emulate)
    return F.mse_loss(out, HR)


def is_trainable(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


def adam_l2_step(p: Params, grads: Params, state: Dict[str, Dict[str, torch.Tensor]],
                 step: int, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8) -> None:
    """torch.optim.Adam (L2-in-grad weight decay, not AdamW) single step, in place
    (train/tactileSR_train.py:212).  ``step`` is the 1-based step count."""
    b1, b2 = betas
    for k, g in grads.items():
        w = p[k]
        g = g + weight_decay * w
        st = state.setdefault(k, {"m": torch.zeros_like(w), "v": torch.zeros_like(w)})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_one_iter(p: Params, state, step: int, LR, HR_raw, lr=1e-3, weight_decay=1e-2,
                   **cfg) -> "Tuple[float, Params]":
    """zero_grad -> backward -> Adam(L2) step (cpu/trainer.py:346-362); BN running stats
    are updated in ``p``.  Returns (loss, grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items() if is_trainable(k)}
    full = dict(p)
    full.update(leaves)
    new_stats: Params = {}
    loss = train_cal_loss(full, LR, HR_raw, training=True, new_stats=new_stats, **cfg)
    gl = torch.autograd.grad(loss, list(leaves.values()))
    grads = {k: g for k, g in zip(leaves.keys(), gl)}
    for k, v in new_stats.items():
        p[k] = v.detach()
    adam_l2_step(p, grads, state, step, lr, weight_decay)
    return float(loss.detach()), grads


# --------------------------------------------------------------------------
# metrics (utility/tools.py:49-81)
# --------------------------------------------------------------------------
def calculation_psnr(a: torch.Tensor, b: torch.Tensor, maxValue: float) -> torch.Tensor:
    """calculationPSNR, utility/tools.py:49-63: the squared-error SUM is divided by
    shape[0]*shape[1] whatever the rank -- for the (1,40,40) tensors eval_func passes
    (train/tactileSR_train.py:89) that is 40, not 1600."""
    mse = ((a - b) ** 2).sum() / (a.shape[0] * a.shape[1])
    return 10 * torch.log10(maxValue ** 2 / mse)


def calculation_ssim(a: torch.Tensor, b: torch.Tensor, C1=0.01 ** 2, C2=0.03 ** 2) -> torch.Tensor:
    """calculationSSIM, utility/tools.py:66-81 (single global window)."""
    mu1, mu2 = a.mean(), b.mean()
    s1 = (a * a).mean() - mu1 * mu1
    s2 = (b * b).mean() - mu2 * mu2
    s12 = (a * b).mean() - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def eval_batch(p: Params, LR, HR_raw, maxValue=250.0, **cfg) -> "Tuple[float, float, float]":
    """One test batch of eval_func (train/tactileSR_train.py:77-94): (mse, mean psnr, mean ssim)."""
    HR = prepare_target(HR_raw, cfg.get("HR_scale_num", 10.0), cfg.get("scale_factor", 10))
    seqsCnt, axisCnt = cfg.get("seqsCnt", 1), cfg.get("axisCnt", 3)
    out = tactilesr_forward(p, LR.type(torch.float32)[:, :seqsCnt * axisCnt],
                            cfg.get("scale_factor", 10), axisCnt, training=False)
    mse = float(F.mse_loss(out, HR))
    ps = sum(float(calculation_psnr(out[i], HR[i], maxValue)) for i in range(out.shape[0]))
    ss = sum(float(calculation_ssim(out[i], HR[i])) for i in range(out.shape[0]))
    return mse, ps / out.shape[0], ss / out.shape[0]


# --------------------------------------------------------------------------
# tPSFNet (model/tPSFNet.py)
# --------------------------------------------------------------------------
def tpsf_state_shapes() -> "Dict[str, Tuple[int, ...]]":
    """MLP_layer.{1,3,5,7}.{weight,bias}; model/tPSFNet.py:26-36."""
    d = {}
    for idx, (i, o) in zip((1, 3, 5, 7), ((48, 256), (256, 1024), (1024, 256), (256, 3))):
        d[f"MLP_layer.{idx}.weight"] = (o, i)
        d[f"MLP_layer.{idx}.bias"] = (o,)
    return d


def tpsf_geometry() -> "Tuple[torch.Tensor, torch.Tensor]":
    """PSF_sdf (1,1,99,99) and LR_masking_sdf (4,4,100,100), each min-max scaled to
    [0,10] (the 16 masks jointly); model/tPSFNet.py:43-55,67-76.  Vectorised fp32
    restatement of the reference's python double loops."""
    u = torch.arange(99, dtype=torch.float32)
    sdf = ((u.view(-1, 1) - 49) ** 2 + (u.view(1, -1) - 49) ** 2) ** 0.5
    psf_sdf = (10 * (sdf - sdf.min()) / (sdf.max() - sdf.min())).view(1, 1, 99, 99)
    xs = torch.arange(100, dtype=torch.float32)
    m = torch.zeros(4, 4, 100, 100)
    for a in range(4):
        for b in range(4):
            cx, cy = 12 + a * 25, 12 + b * 25
            m[a, b] = ((xs.view(-1, 1) - cx) ** 2 + (xs.view(1, -1) - cy) ** 2) ** 0.5
    m = 10 * (m - m.min()) / (m.max() - m.min())
    return psf_sdf, m


def tpsf_mlp(p: Params, x: torch.Tensor) -> torch.Tensor:
    """Flatten -> Linear/ReLU x3 -> Linear -> Softplus; model/tPSFNet.py:26-36."""
    h = x.flatten(1)
    h = F.relu(F.linear(h, p["MLP_layer.1.weight"], p["MLP_layer.1.bias"]))
    h = F.relu(F.linear(h, p["MLP_layer.3.weight"], p["MLP_layer.3.bias"]))
    h = F.relu(F.linear(h, p["MLP_layer.5.weight"], p["MLP_layer.5.bias"]))
    return F.softplus(F.linear(h, p["MLP_layer.7.weight"], p["MLP_layer.7.bias"]))


def tpsf_forward_from_ab(ab: torch.Tensor, depth: torch.Tensor, geom=None):
    """The PSF forward model for given (alpha, beta, gamma) rows `ab` (B,3) and depth (B,1,100,100) or (B,100,100),
    model/tPSFNet.py:78-100,102-127,129-141: per sample psf = a*exp(-sdf^2/b^2); HR = conv2d(ZeroPad48(depth), psf,
    padding=1); plateau fill (HR[mask] = max(HR outside mask)); 16 Gaussian-masked sums * 1e-4.  Runs in the dtype
    of `ab` (tests use float64 as the yardstick)."""
    psf_sdf, mask_sdf = geom if geom is not None else tpsf_geometry()
    psf_sdf, mask_sdf = psf_sdf.to(ab.dtype), mask_sdf.to(ab.dtype)
    if depth.dim() == 3:
        depth = depth.unsqueeze(1)
    B = ab.shape[0]
    HRs, LRs, psfs = [], [], []
    for i in range(B):
        psf = ab[i, 0] * torch.exp(-psf_sdf ** 2 / (ab[i, 1] ** 2))
        d = depth[i:i + 1]
        dmask = d > (d.max() - 1e-3)
        HR = F.conv2d(F.pad(d, (48, 48, 48, 48)), psf, padding=1)
        tmp = HR.detach().clone()
        tmp[dmask] = 0
        fill = tmp.max()
        HR = torch.where(dmask, fill, HR)
        m = torch.exp(-mask_sdf ** 2 / ab[i, 2])
        m = (m - m.min()) / (m.max() - m.min())
        LRd = (HR[0, 0].unsqueeze(0).unsqueeze(0) * m).sum(dim=(2, 3)) * 1e-4
        HRs.append(HR)
        LRs.append(LRd.view(1, 1, 4, 4))
        psfs.append(psf)
    return torch.cat(HRs), torch.cat(LRs), torch.cat(psfs)


def tpsf_forward(p: Params, x: torch.Tensor, depth: torch.Tensor, geom=None):
    """tPSFNet.forward, model/tPSFNet.py:102-127: MLP -> (alpha, beta, gamma) -> the PSF forward model above."""
    assert x.shape[0] == depth.shape[0], "Batch size of LR tactile and depth should be the same!"
    ab = tpsf_mlp(p, x)
    HR, LRd, psf = tpsf_forward_from_ab(ab, depth, geom)
    return HR, LRd, psf, ab.view(x.shape[0], 1, 3)


def tpsf_train_cal_loss(p: Params, LR_raw: torch.Tensor, depth: torch.Tensor, scale_num=100.0,
                        geom=None) -> torch.Tensor:
    """Trainer_tPSF.train_cal_loss, train/tPSFNet_train.py:180-190."""
    LR = LR_raw.type(torch.float32) / scale_num
    d = depth.type(torch.float32).unsqueeze(1)
    _, LRd, _, _ = tpsf_forward(p, LR, d, geom)
    return F.mse_loss(LR[:, 2:3], LRd)


# --------------------------------------------------------------------------
# fp64 numpy "truth" helpers for single ops (small cases; used to bound fp32 error)
# --------------------------------------------------------------------------
def conv2d_f64(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], pad: int) -> np.ndarray:
    """Direct cross-correlation in float64: x (B,C,H,W), w (O,C,kh,kw)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    Ho, Wo = H + 2 * pad - kh + 1, W + 2 * pad - kw + 1
    out = np.zeros((B, O, Ho, Wo), np.float64)
    for i in range(kh):
        for j in range(kw):
            out += np.einsum("bchw,oc->bohw", xp[:, :, i:i + Ho, j:j + Wo], w[:, :, i, j])
    if bias is not None:
        out += np.asarray(bias, np.float64).reshape(1, -1, 1, 1)
    return out
