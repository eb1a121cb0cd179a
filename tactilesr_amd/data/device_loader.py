"""Device-resident batch loader for SR training / evaluation.

The reference feeds ``Trainer_tactileSR`` from ``DataLoader(TactileSRDataset(file), batch_size, shuffle)``
(train/tactileSR_train.py:58-59, utility/load_tactile_dataset.py:39-47): every step moves a host batch to the GPU
(train/tactileSR_train.py:43), 40 KB of HR per sample.  At thousands of samples per second per GPU that copy and the
per-sample ``.item()`` unboxing are the step's host bottleneck, so this loader keeps the whole (LR, HR) set on the
device and yields index-gathered batches with the same ``(LR, HR)`` tuple protocol -- ``train_one_iter`` /
``eval_func`` take it unchanged.  Shuffling follows ``torch.randperm`` under its own generator (a
``RandomSampler``-style fresh permutation per epoch; the order is NOT the reference DataLoader's, whose sampler
draws from the global CPU generator).
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch


class DeviceSRLoader:
    def __init__(self, LR: torch.Tensor, HR: torch.Tensor, batch_size: int, shuffle: bool = False,
                 drop_last: bool = False, seed: Optional[int] = None, device="cuda",
                 rank: int = 0, world_size: int = 1):
        assert LR.shape[0] == HR.shape[0], "LR and HR must hold the same number of samples"
        dev = torch.device(device)
        # shard of the set for this rank (data-parallel training: one process per GPU).  EVERY rank gets the same
        # number of samples -- ceil(n/world), the tail wrapping around to the head like DistributedSampler's padding --
        # so every rank runs the same number of batches per epoch and nobody is left alone in an all-reduce.
        n = LR.shape[0]
        if world_size > 1:
            from ..ddp import equal_shard
            idx = torch.as_tensor(equal_shard(n, rank, world_size), dtype=torch.long)
            LR, HR = LR.index_select(0, idx.to(LR.device)), HR.index_select(0, idx.to(HR.device))
        self.LR = LR.to(dev).contiguous()
        self.HR = HR.to(dev).contiguous()
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(0 if seed is None else int(seed))
        self.epoch = 0

    @classmethod
    def from_entries(cls, entries: List[list], batch_size: int, **kw) -> "DeviceSRLoader":
        """From the generator's in-memory entries (tactilesr_amd.data.depth2tactile.synthesize)."""
        LR = torch.stack([torch.as_tensor(e[0]["LR"]) for e in entries])
        HR = torch.stack([torch.as_tensor(e[0]["HR"]) for e in entries])
        return cls(LR, HR, batch_size, **kw)

    @classmethod
    def from_file(cls, path: str, batch_size: int, **kw) -> "DeviceSRLoader":
        """From a dataset file in the reference's on-disk format (an object ``.npy`` of ``{'LR','HR',...}`` dicts)
        written by ``save_dataset``.  Object arrays are pickles: only load files you wrote."""
        arr = np.load(path, allow_pickle=True)
        entries = [[arr[i].item() if arr[i].shape == () else arr[i, 0]] for i in range(len(arr))]
        return cls.from_entries(entries, batch_size, **kw)

    def __len__(self) -> int:
        n = self.LR.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        n = self.LR.shape[0]
        order = torch.randperm(n, generator=self._gen, device=self.LR.device) if self.shuffle else None
        self.epoch += 1
        for i in range(len(self)):
            lo, hi = i * self.batch_size, min(n, (i + 1) * self.batch_size)
            if order is None:
                yield self.LR[lo:hi], self.HR[lo:hi]
            else:
                idx = order[lo:hi]
                yield self.LR.index_select(0, idx), self.HR.index_select(0, idx)
