"""Batched generator of the multi-frame (Seqs) SR dataset: the offline step between tPSFNet training and
tactileSRSeqs training (reference data/SeqsDataset/seqsDepth2Tactile.py:20-107).

What the reference does, per (contact, translation, tap-sequence sample): picks seven taps of the same contact at
rotations 0..30 degrees -- for 0..25 degrees the LAST sample of that rotation's tap sequence, for 30 degrees the
sample ``seqs_idx`` -- divides their LR readings by ``scale_num``, runs tPSFNet at BATCH 1 on the 30-degree tap only,
stacks the seven LR frames NEWEST FIRST (30, 25, ..., 0 degrees) into ``LR (21,4,4)`` and stores
``{'LR', 'depth' (the 30-degree depth, (1,100,100)), 'HR' (tPSFNet's HR of the 30-degree tap)}``; translation 0 goes
to the test split, translation 1 to validation, the rest to train (:42-98).

Here the index arithmetic is a table built once (``seqs_index_table``), the seven taps are gathered with one
``index_select`` and ALL 30-degree taps go through the batched HIP tPSFNet forward in chunks -- no python loop over
samples touches the GPU.  The on-disk format is the reference's (an object ``.npy`` of one-element lists of dicts,
utility/load_tactile_dataset.py:52-57).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

ROT_SLOTS = 9            # rotation slots per translation in the flat tap order (the generator uses slots 0..6)
TRANS_SLOTS = 9          # translations per contact in the flat tap order (81 tap sequences per contact)
N_FRAMES = 7


def seqs_index_table(n_contacts: int = 18, n_trans: int = 9, sample_cnt: int = 16) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx, trans): ``idx`` (n_items, 7) int64 flat tap indices, column 0 = the 30-degree tap (newest), column 6 =
    the 0-degree tap; ``trans`` (n_items,) the translation index that decides the split.  Flat tap order is the
    reference's tPSFNetDataSet order: ((contact * 9 + trans) * 9 + rotation) * sample_cnt + sample (:50-56);
    ``n_trans`` <= 9 only limits how many translations are generated, the contact stride stays 81 sequences."""
    assert 1 <= n_trans <= TRANS_SLOTS
    c = torch.arange(n_contacts).view(-1, 1, 1)
    t = torch.arange(n_trans).view(1, -1, 1)
    s = torch.arange(sample_cnt).view(1, 1, -1)
    base = (c * TRANS_SLOTS + t) * ROT_SLOTS * sample_cnt                   # first tap of (contact, trans)
    cols = [base + 6 * sample_cnt + s]                                      # 30 degrees: sample seqs_idx
    for rot in (5, 4, 3, 2, 1, 0):                                          # 25 .. 0 degrees: the last sample
        cols.append((base + rot * sample_cnt + (sample_cnt - 1)).expand(n_contacts, n_trans, sample_cnt))
    idx = torch.stack([col.expand(n_contacts, n_trans, sample_cnt) for col in cols], dim=-1).reshape(-1, N_FRAMES)
    trans = t.expand(n_contacts, n_trans, sample_cnt).reshape(-1)
    return idx.long(), trans.long()


@torch.no_grad()
def synthesize_seqs(tpsf_model, LR_raw: torch.Tensor, depth: torch.Tensor, n_contacts: int = 18, n_trans: int = 9,
                    sample_cnt: int = 16, scale_num: float = 100.0, batch_size: int = 4096,
                    validation_idx=(1,), test_idx=(0,)) -> Dict[str, List[list]]:
    """LR_raw (N,3,4,4) sensor units and depth (N,100,100) in the flat tap order above ->
    {'train' | 'validation' | 'test': list of [dict] entries} (reference :82-98)."""
    idx, trans = seqs_index_table(n_contacts, n_trans, sample_cnt)
    assert int(idx.max()) < LR_raw.shape[0] == depth.shape[0], "tap arrays are shorter than the index table needs"
    dev = next(tpsf_model.parameters()).device
    tpsf_model.eval()
    LR_all = LR_raw.type(torch.float32) / scale_num
    n = idx.shape[0]
    frames = LR_all.index_select(0, idx.reshape(-1)).view(n, 3 * N_FRAMES, 4, 4)        # newest first
    newest = idx[:, 0]
    HR = torch.empty(n, 1, 100, 100)
    for i in range(0, n, batch_size):
        sel = newest[i:i + batch_size]
        d = depth.index_select(0, sel).to(dev).type(torch.float32).unsqueeze(1)
        hr, _, _, _ = tpsf_model(LR_all.index_select(0, sel).to(dev), d)
        HR[i:i + batch_size] = hr.cpu()
    out: Dict[str, List[list]] = {"train": [], "validation": [], "test": []}
    d30 = depth.index_select(0, newest).type(depth.dtype).unsqueeze(1)
    for k in range(n):
        t = int(trans[k])
        split = "validation" if t in validation_idx else ("test" if t in test_idx else "train")
        out[split].append([{"LR": frames[k].clone(), "depth": d30[k].clone(), "HR": HR[k].clone()}])
    return out


def save_seqs_dataset(path: str, entries: List[list]) -> None:
    arr = np.empty((len(entries), 1), dtype=object)
    for i, e in enumerate(entries):
        arr[i, 0] = e[0]
    np.save(path, arr, allow_pickle=True)
