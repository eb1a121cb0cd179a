"""Checkpoint files in the layout the reference's trainer writes and its consumers read
(reference cpu/trainer.py:394-421 save, :440-475 load; consumers of ``checkpoint['model']``:
train/tactileSRSeqs_train.py:45-53, data/SRdataset/depth2tactile.py:73-74).

Keys: ``num_gpus``, ``model``, ``optimizer``, ``lr_scheduler``, ``metric_storage``, ``epoch`` | ``iter``
(+ ``hooks``, ``grad_scaler`` when present).  The module and optimizer state dicts are plain torch ones
(same key names as the reference's), so a file written by either side loads on the other.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def save_checkpoint(path: str, model, optimizer=None, lr_scheduler=None, epoch: Optional[int] = None,
                    iteration: Optional[int] = None, num_gpus: Optional[int] = None, metric_storage=None,
                    hooks: Optional[dict] = None) -> None:
    module = model.module if hasattr(model, "module") else model     # DDP unwrap (cpu/trainer.py:172-176)
    if num_gpus is None:
        num_gpus = torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                          and torch.distributed.is_initialized()) else 1
    data = {"num_gpus": num_gpus, "model": module.state_dict(),
            "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None,
            "metric_storage": metric_storage}
    data.update({"epoch": epoch} if iteration is None else {"iter": iteration})
    if hooks:
        data["hooks"] = hooks
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(data, path)
    latest = os.path.join(os.path.dirname(os.path.abspath(path)), "latest.pth")
    if os.path.lexists(latest):
        os.remove(latest)
    os.symlink(os.path.basename(path), latest)


def load_checkpoint(path: str, model, optimizer=None, lr_scheduler=None, num_gpus: Optional[int] = None,
                    trusted: bool = False) -> dict:
    """Restore model (strict=False like the reference), optimizer and scheduler; returns the dict.
    Tensors-only files load with ``weights_only=True``; a file that also carries pickled python objects
    (the reference stores its MetricStorage instance) needs ``trusted=True`` -- only for files you wrote."""
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        if not trusted:
            raise
        ck = torch.load(path, map_location="cpu", weights_only=False)
    if num_gpus is not None:
        assert ck["num_gpus"] == num_gpus, (f"You are trying to load a checkpoint trained with {ck['num_gpus']} "
                                            f"GPUs, but currently only have {num_gpus} GPUs.")
    module = model.module if hasattr(model, "module") else model
    module.load_state_dict(ck["model"], strict=False)
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    if lr_scheduler is not None and ck.get("lr_scheduler") is not None:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
    return ck


def model_param_init(seqs_model, single_frame_state_dict, make_single_model):
    """The Seqs trainer's weight transplant (train/tactileSRSeqs_train.py:43-59): build the single-frame
    model, load its checkpointed weights, and REPLACE the two feature-extraction submodules of the
    multi-frame model by it.  As in the reference, an optimizer created before this call keeps pointing at
    the discarded modules, so the transplanted blocks stay frozen (only BN running stats move)."""
    single = make_single_model()
    single.load_state_dict(single_frame_state_dict, strict=False)
    single = single.to(next(seqs_model.parameters()).device)
    seqs_model.patternFeatureExtra_layer = single.patternFeatureExtra_layer
    seqs_model.forceFeatureExtra_layer = single.forceFeatureExtra_layer
    return seqs_model
