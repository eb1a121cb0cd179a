"""Checkpoint files in the layout the reference's trainer writes and its consumers read
(reference cpu/trainer.py:394-421 save, :440-475 load; consumers of ``checkpoint['model']``:
train/tactileSRSeqs_train.py:45-53, data/SRdataset/depth2tactile.py:73-74).

Keys: ``num_gpus``, ``model``, ``optimizer``, ``lr_scheduler``, ``metric_storage``, ``epoch`` | ``iter``
(+ ``hooks``, ``grad_scaler`` when present).  The module and optimizer state dicts are plain torch ones (same key
names as the reference's).  Both directions are covered: ``load_checkpoint`` reads a dict laid out as the reference
writes it (its warm-up scheduler's key layout, a pickled metric-store object, ``hooks``, ``grad_scaler``), and
``save_checkpoint`` writes a ``metric_storage`` object with the protocol the reference's hooks call
(``train/metrics.py``) and, on request, the scheduler state in the reference's layout.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def save_checkpoint(path: str, model, optimizer=None, lr_scheduler=None, epoch: Optional[int] = None,
                    iteration: Optional[int] = None, num_gpus: Optional[int] = None, metric_storage=None,
                    hooks: Optional[dict] = None, grad_scaler=None, reference_layout: bool = False) -> None:
    """Write the dict of reference cpu/trainer.py:401-411.  ``metric_storage`` defaults to an empty
    ``tactilesr_amd.train.metrics.MetricStorage`` -- the reference installs whatever object sits under that key as its
    live store on resume (cpu/trainer.py:469), so ``None`` would break it.  ``grad_scaler`` (a GradScaler or its state
    dict) is written only when given: the reference asserts that the key is present iff AMP is on (:477-478).
    ``reference_layout=True`` writes the LR-scheduler state in the key layout of the reference's warm-up class."""
    module = model.module if hasattr(model, "module") else model     # DDP unwrap (cpu/trainer.py:172-176)
    if num_gpus is None:
        num_gpus = torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                          and torch.distributed.is_initialized()) else 1
    if metric_storage is None:
        from .metrics import MetricStorage
        metric_storage = MetricStorage()
    sched_state = None
    if lr_scheduler is not None:
        sched_state = (lr_scheduler.reference_state_dict() if reference_layout and hasattr(lr_scheduler, "reference_state_dict")
                       else lr_scheduler.state_dict())
    data = {"num_gpus": num_gpus, "model": module.state_dict(),
            "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "lr_scheduler": sched_state, "metric_storage": metric_storage}
    data.update({"epoch": epoch} if iteration is None else {"iter": iteration})
    if hooks:
        data["hooks"] = hooks
    if grad_scaler is not None:
        data["grad_scaler"] = grad_scaler.state_dict() if hasattr(grad_scaler, "state_dict") else grad_scaler
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(data, path)
    latest = os.path.join(os.path.dirname(os.path.abspath(path)), "latest.pth")
    if os.path.lexists(latest):
        os.remove(latest)
    os.symlink(os.path.basename(path), latest)


def load_checkpoint(path: str, model, optimizer=None, lr_scheduler=None, num_gpus: Optional[int] = None,
                    trusted: bool = False, grad_scaler=None, hooks: Optional[dict] = None,
                    epoch_len: Optional[int] = None) -> dict:
    """Restore model (strict=False like the reference), optimizer, scheduler, grad scaler and hook states, in the
    order of reference cpu/trainer.py:440-498; returns the checkpoint dict with ``start_iter`` added
    (``(epoch+1)*epoch_len`` or ``iter+1``, :452-457).

    ``hooks``: {class name: object with load_state_dict}; states without a taker and takers without a state are
    reported in ``ck['hooks_missing']`` / ``ck['hooks_unexpected']`` (the reference only warns).  AMP consistency
    (:477-478): a ``grad_scaler`` argument demands a ``grad_scaler`` entry and vice versa.
    Files written by this package load with the restricted unpickler (``weights_only=True`` + this package's metric
    store classes allow-listed); a file that carries other pickled python objects (every reference-written file does:
    its MetricStorage instance) needs ``trusted=True`` -- only for files whose origin you trust."""
    from . import metrics as _m
    try:   # this package's own metric-store classes are allow-listed for the restricted unpickler
        with torch.serialization.safe_globals([_m.MetricStorage, _m.Series, _m._rebuild]):
            ck = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        if not trusted:
            raise
        ck = torch.load(path, map_location="cpu", weights_only=False)
    if num_gpus is not None:
        assert ck["num_gpus"] == num_gpus, (f"You are trying to load a checkpoint trained with {ck['num_gpus']} "
                                            f"GPUs, but currently only have {num_gpus} GPUs.")
    if "epoch" in ck:
        ck["start_iter"] = (ck["epoch"] + 1) * epoch_len if epoch_len is not None else None
    else:
        ck["start_iter"] = ck["iter"] + 1
    module = model.module if hasattr(model, "module") else model
    incompatible = module.load_state_dict(ck["model"], strict=False)
    ck["missing_keys"], ck["unexpected_keys"] = list(incompatible.missing_keys), list(incompatible.unexpected_keys)
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    if lr_scheduler is not None and ck.get("lr_scheduler") is not None:
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
    assert (grad_scaler is not None) == ("grad_scaler" in ck), \
        "Found inconsistent AMP training setting when loading checkpoint."
    if grad_scaler is not None:
        grad_scaler.load_state_dict(ck["grad_scaler"])
    states = ck.get("hooks", {})
    takers = hooks or {}
    ck["hooks_missing"] = [n for n in takers if n not in states]
    ck["hooks_unexpected"] = [n for n in states if n not in takers]
    for name, st in states.items():
        if name in takers:
            takers[name].load_state_dict(st)
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
