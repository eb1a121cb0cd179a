"""Gradient parity on the activation pattern the device took (helper of the -m gpu training tests).

Why: the network's loss is piecewise linear-ish in every ReLU, so its gradient is DISCONTINUOUS where a
pre-activation is zero to rounding.  Two faithful fp32 forwards (the reference's CPU run, this HIP path, an fp64
run) disagree on the sign of a handful of such pre-activations per batch, and each disagreement moves whole rows of
weight-gradient entries by ~1e-3 of the tensor max.  A max-norm bar against a gradient evaluated on a DIFFERENT
activation pattern therefore cannot be tight, and a loose one hides real kernel bugs.

So the tests do both halves explicitly:
  1. pattern check -- the HIP masks equal the fp64 oracle's own masks except at elements whose fp64 pre-activation
     is within `flip_tol` (relative to that tensor's max) of zero; the count of such flips is reported and bounded;
  2. gradient check -- the fp64 oracle gradient is re-evaluated ON the HIP masks (oracle.ReluTap), and every
     parameter gradient of the HIP path must match it in max-norm (relative to the tensor's max) to `tol`.
"""
import torch
import torch.nn.functional as F

from oracle import tactilesr_oracle as O


def oracle_grads(sd, LR, HR, scale_factor=10, dtype=torch.float64, masks=None, record=False):
    """loss, {param: grad}, new BN stats and (optionally) the ReLU pre-activations of one train-mode forward +
    backward of the CPU oracle in `dtype`, with the ReLU pattern optionally forced to `masks`."""
    leaves = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd.items() if O.is_trainable(k)}
    full = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    full.update(leaves)
    tap = O.ReluTap(masks=masks, record=record) if (masks is not None or record) else None
    ns = {}
    out = O.tactilesr_forward(full, LR.to(dtype), scale_factor=scale_factor, training=True, new_stats=ns, tap=tap)
    loss = F.mse_loss(out, HR.to(dtype))
    gl = torch.autograd.grad(loss, list(leaves.values()))
    return float(loss), dict(zip(leaves, gl)), ns, (tap.pre if tap is not None else None)


def check_pattern(hip_masks, pre64, flip_tol=1e-5, max_flip_frac=2e-5):
    """HIP activation pattern vs the fp64 oracle's: every disagreement must sit at a pre-activation that is zero to
    rounding.  Returns the number of flipped elements."""
    assert set(hip_masks) == set(pre64), (sorted(set(hip_masks) ^ set(pre64)))
    flips = total = 0
    for name, pre in pre64.items():
        hm = hip_masks[name].cpu()
        assert hm.shape == pre.shape, (name, hm.shape, pre.shape)
        diff = hm != (pre > 0)
        n = int(diff.sum())
        total += pre.numel()
        if n:
            worst = float(pre[diff].abs().max() / pre.abs().max())
            assert worst < flip_tol, f"{name}: mask differs where the fp64 pre-activation is {worst:.2e} of max"
            flips += n
    assert flips <= max(4, max_flip_frac * total), (flips, total)
    return flips


def check_grads(named_grads, g64m, tol=2e-5, zero_tol=1e-4):
    """max-norm bar on EVERY parameter gradient against the fp64 gradient evaluated on the HIP masks."""
    worst = (0.0, None)
    bad = []
    for k, ref in g64m.items():
        got = named_grads[k].detach().cpu().double()
        den = float(ref.abs().max())
        if den < 1e-6:            # conv bias in front of a train-mode BN: the exact gradient is 0
            assert float(got.abs().max()) < zero_tol, k
            continue
        e = float((got - ref).abs().max()) / den
        if e > worst[0]:
            worst = (e, k)
        if not e <= tol:
            bad.append((k, e))
    assert not bad, bad
    return worst
