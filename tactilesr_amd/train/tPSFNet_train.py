"""Mirror of Trainer_tPSF.train_cal_loss (reference train/tPSFNet_train.py:180-190) on the HIP path."""
from __future__ import annotations

import torch

from .. import functional as Fh


def train_cal_loss(model, batch, scale_num=100.0):
    device = next(model.parameters()).device
    LR, depth = batch
    LR, depth = LR.to(device), depth.to(device)
    LR, depth = LR.type(torch.float32) / scale_num, depth.type(torch.float32)
    depth = depth.unsqueeze(1)
    HR_tactile, LR_tactile_degrade, ret_psf, ret_alphaBeta = model(LR, depth)
    loss = Fh.mse_loss(LR_tactile_degrade, LR[:, 2:3].contiguous())
    return loss, {"total_loss": loss}
