"""ROCm-safe stand-in for the reference's ``config/default.py`` (reference config/default.py:6-104).

Same three dictionaries with the same keys and values -- ``tPSFNet_config``, ``tactileSR_config``,
``tactileSeqs_config`` -- and the same module globals ``root_path`` and ``device`` that the scripts import
(train/tactileSR_train.py:20, train/tPSFNet_train.py:21, data/*/…py).  Differences, both forced by the platform:

* ``device`` -- the reference shells out to ``nvidia-smi`` at import time and then sets ``CUDA_VISIBLE_DEVICES='0, 1'``
  (config/default.py:101-104); here it is ``cuda:<LOCAL_RANK>`` (one process per GPU: torchrun's LOCAL_RANK, 0 when
  absent) resolved lazily, no subprocess, no environment mutation.  Importing this module never touches the GPU.
* ``root_path`` -- the reference hard-wires ``/code`` (config/default.py:6); here ``$TACTILESR_ROOT`` or the current
  working directory.
"""
from __future__ import annotations

import os


def _root() -> str:
    return os.environ.get("TACTILESR_ROOT", os.getcwd())


root_path = _root()


def _p(*parts: str) -> str:
    return os.path.join(root_path, *parts)


common_config = dict(root_path=root_path, random_seed=42, deterministic=False, scale_num=100)

# tPSFNet trainer (reference config/default.py:17-41)
tPSFNet_config = dict(
    common_config,
    train_batch_size=256, test_batch_size=8,
    gama=1.4, perception_scale=None, loss_scale=1e-1,
    lr=1e-4, lr_scheduler_step_size=1, lr_scheduler_gamma=0.8, weight_decay=1e-5,
    checkpoint_period=1, epochs=51, sample_cnt=32,
    dataset_dir=_p("data/rotateDataset"), save_dir=_p("pth/tPSFNet_no_aug"), is_aug_data=False,
    inference_test=True, inference_index=36, inference_seqs_length=64,
    test_dataset_dir_1=_p("data/rotateDataset/I.npy"), test_dataset_dir_2=_p("data/rotateDataset/P.npy"),
)

# single-frame SR trainer (reference config/default.py:45-77)
tactileSR_config = dict(
    common_config,
    train_batch_size=32, test_batch_size=8,
    lr=1e-3, weight_decay=1e-2, lr_scheduler_step_size=2, lr_scheduler_gamma=0.8,
    checkpoint_period=1, epochs=51,
    HR_scale_num=10,
    sensorMaxVaule_factor=250,     # (sic) passed raw as PSNR maxValue: train/tactileSR_train.py:70,89
    warmup_t=2000, warmup_by_epoch=True, warmup_mode="auto", warmup_init_lr=1e-5, warmup_factor=1e-4,
    scale_factor=10, seqsCnt=1, axisCnt=3, patternFeatureExtraLayerCnt=6, forceFeatureExtraLayerCnt=1,
    inference_test=True,
    save_dir=_p("pth/tactileSR_single"),
    train_dataset_dir=_p("data/SRdataset/SRdataset_train.npy"),
    test_dataset_dir=_p("data/SRdataset/SRdataset_test.npy"),
    val_dataset_dir=_p("data/SRdataset/SRdataset_validation.npy"),
)

# multi-frame (Seqs) SR trainer: the single-frame dict with these overrides (reference config/default.py:80-96)
tactileSeqs_config = dict(
    tactileSR_config,
    seqsCnt=7, axisCnt=3,
    lr=1e-4, weight_decay=1e-2, epochs=51,
    load_checkpoint_dir=_p("pth/tactileSR_single/checkpoints/epoch_50.pth"),
    save_dir=_p("pth/tactileSeqs_seq_7"),
    train_dataset_dir=_p("data/SeqsDataset/SRdataset_train_32.npy"),
    test_dataset_dir=_p("data/SeqsDataset/SRdataset_test_32.npy"),
    val_dataset_dir=_p("data/SeqsDataset/SRdataset_validation_32.npy"),
)


def pick_device():
    """``torch.device`` of this process's GPU: LOCAL_RANK-th visible device.  Raises if no ROCm device exists -- the
    package has no CPU path."""
    import torch
    if not torch.cuda.is_available():
        from .._lib import TactileSRHipError
        raise TactileSRHipError("no ROCm device visible: tactilesr_amd has no CPU fallback")
    n = torch.cuda.device_count()
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % n)


def __getattr__(name):       # `from tactilesr_amd.config.default import device` resolves lazily (PEP 562)
    if name == "device":
        return pick_device()
    raise AttributeError(name)
