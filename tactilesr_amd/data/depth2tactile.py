"""Batched HR-target synthesis with a trained tPSFNet: the offline dataset generator that sits between
tPSFNet training and SR training (reference data/SRdataset/depth2tactile.py:104-160).  The reference runs
the model at batch 1 inside a python loop; here the whole set goes through the batched HIP forward.

On-disk format is the reference's: ``np.save`` of a list of one-element lists of dicts
``{'LR' (3,4,4), 'depth' (1,100,100), 'HR' (1,100,100), 'LR_degrade' (1,4,4), 'alphaBeta' (3,)}`` of CPU torch
tensors -> an object array read back by ``TactileSRDataset`` as ``np.load(..., allow_pickle=True)[i].item()``
(utility/load_tactile_dataset.py:39-47).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch


@torch.no_grad()
def synthesize(tpsf_model, LR_raw: torch.Tensor, depth: torch.Tensor, scale_num: float = 100.0,
               batch_size: int = 4096) -> List[list]:
    """LR_raw (N,3,4,4) sensor units, depth (N,100,100) -> list of [dict] entries (reference :107-119)."""
    dev = next(tpsf_model.parameters()).device
    tpsf_model.eval()
    out = []
    for i in range(0, LR_raw.shape[0], batch_size):
        LR = LR_raw[i:i + batch_size].to(dev).type(torch.float32) / scale_num
        d = depth[i:i + batch_size].to(dev).type(torch.float32).unsqueeze(1)
        HR, LRd, _, ab = tpsf_model(LR, d)
        LR, d, HR, LRd, ab = LR.cpu(), d.cpu(), HR.cpu(), LRd.cpu(), ab.cpu()
        for k in range(LR.shape[0]):
            out.append([{"LR": LR[k].clone(), "depth": d[k].clone(), "HR": HR[k].clone(),
                         "LR_degrade": LRd[k].clone(), "alphaBeta": ab[k, 0].clone()}])
    return out


def save_dataset(path: str, entries: List[list]) -> None:
    arr = np.empty((len(entries), 1), dtype=object)
    for i, e in enumerate(entries):
        arr[i, 0] = e[0]
    np.save(path, arr, allow_pickle=True)
