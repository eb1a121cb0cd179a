"""Host-side mirror of the reference's SR trainer glue (train/tactileSR_train.py:29-101) on the
HIP path.  The reference subclasses its vendored ``cpu.Trainer`` (out of scope here); what the hot
path needs from it are the three functions below with the same argument meaning:

  * ``train_cal_loss(model, batch, config)``  <->  Trainer_tactileSR.train_cal_loss   (:41-51)
  * ``train_one_iter(model, optimizer, batch, config)``  <->  Trainer.train_one_iter  (cpu/trainer.py:346-362)
  * ``eval_func(model, test_loader, config)``  <->  eval_func (:66-101), returning the three averages
    the reference only logs.
"""
from __future__ import annotations

import torch

from .. import functional as Fh


def default_config():
    """The keys of the reference's tactileSR_config that the path reads (config/default.py:45-72)."""
    return dict(HR_scale_num=10, sensorMaxVaule_factor=250, scale_factor=10, seqsCnt=1, axisCnt=3,
                lr=1e-3, weight_decay=1e-2, train_batch_size=32, test_batch_size=8)


def _prep(batch, config, device):
    LR, HR = batch
    LR, HR = LR.to(device), HR.to(device)
    HR = Fh.prepare_target(HR, config["HR_scale_num"], config["scale_factor"])
    LR = LR.type(torch.float32)[:, :config["seqsCnt"] * config["axisCnt"]]
    return LR, HR


def train_cal_loss(model, batch, config):
    device = next(model.parameters()).device
    LR, HR = _prep(batch, config, device)
    out = model(LR)
    loss = Fh.mse_loss(out, HR)
    return loss, {"total_loss": loss}


def train_one_iter(model, optimizer, batch, config, grad_sync=None, check_finite=False, cur_iter=None):
    """zero_grad -> backward -> step, the order of cpu/trainer.py:352-361.  ``grad_sync`` (optional)
    is called between backward and step (data-parallel gradient averaging: tactilesr_amd.ddp).

    ``check_finite=True`` reproduces the reference trainer's per-iteration failure check
    (``Trainer._log_iter_metrics``, cpu/trainer.py:259,280-284): the loss is read back (a host sync, like the
    reference's ``loss.detach().cpu().item()``) and a NaN / Inf raises ``FloatingPointError`` with the reference's
    message.  Off by default so that a step enqueues without a device round trip."""
    if grad_sync is not None and hasattr(grad_sync, "pre_forward"):
        grad_sync.pre_forward()          # broadcast_buffers=True: rank 0's BatchNorm statistics, like torch DDP
    losses, loss_dict = train_cal_loss(model, batch, config)
    optimizer.zero_grad()
    losses.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    if check_finite:
        import math
        value = float(losses.detach())
        if not math.isfinite(value):
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={cur_iter}! "
                                     f"loss_dict={ {k: float(v.detach()) for k, v in loss_dict.items()} }.")
    return loss_dict


@torch.no_grad()
def eval_func(model, test_loader, config):
    device = next(model.parameters()).device
    model.eval()
    tot_loss = tot_ssim = tot_psnr = 0.0
    n = 0
    for batch in test_loader:
        LR, HR = _prep(batch, config, device)
        out = model(LR)
        tot_loss += float(Fh.mse_loss(out, HR))
        ps, ss = Fh.psnr_ssim(out, HR, config["sensorMaxVaule_factor"], reference_quirk=True)
        tot_psnr += float(ps.mean())
        tot_ssim += float(ss.mean())
        n += 1
    return tot_loss / n, tot_ssim / n, tot_psnr / n
