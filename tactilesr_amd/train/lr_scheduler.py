"""Warm-up wrapper around a torch LR scheduler with the behaviour of the reference's
``cpu.lr_scheduler.LRWarmupScheduler`` (reference cpu/lr_scheduler.py:41-166), which
``Trainer.__init__`` builds from the ``warmup_*`` arguments of train/tactileSR_train.py:215-228.

Behaviour worth knowing (and reproduced): constructing the wrapper ADVANCES the wrapped
scheduler while it tabulates the no-warm-up learning rates; in "auto" mode ``warmup_init_lr`` is
ignored -- the ramp goes linearly from ``base_lr*warmup_factor`` to the tabulated rate at the end
of warm-up; with an epoch-based scheduler and iteration-based warm-up the wrapped scheduler is not
stepped at epoch ends that fall inside the warm-up.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

from torch.optim.lr_scheduler import ReduceLROnPlateau


class LRWarmupScheduler:
    MODES = ("fix", "auto", "factor")

    def __init__(self, torch_scheduler, by_epoch: bool = True, epoch_len: Optional[int] = None,
                 warmup_t: int = 0, warmup_by_epoch: bool = False, warmup_mode: str = "fix",
                 warmup_init_lr: Optional[float] = None, warmup_factor: Optional[float] = None):
        self.torch_scheduler = torch_scheduler
        self.by_epoch, self.epoch_len = by_epoch, epoch_len
        self.warmup_t, self.warmup_by_epoch = warmup_t, warmup_by_epoch
        self.warmup_mode, self.warmup_init_lr, self.warmup_factor = warmup_mode, warmup_init_lr, warmup_factor
        assert by_epoch or not warmup_by_epoch
        assert not (by_epoch and warmup_t and not warmup_by_epoch) or epoch_len is not None
        assert by_epoch or not self._is_plateau
        self.param_groups = torch_scheduler.optimizer.param_groups
        self.base_lrs = [g["lr"] for g in self.param_groups]
        self.last_iter = self.last_epoch = 0
        self.in_iter_warmup = False
        if not warmup_t:
            return
        # table of the rates the wrapped scheduler would give at t = 0..n (this steps it n times)
        n = warmup_t // epoch_len if (by_epoch and not warmup_by_epoch) else warmup_t
        table = [list(self.base_lrs)]
        for _ in range(n):
            if self._is_plateau:
                table.append(list(self.base_lrs))
            else:
                torch_scheduler.step()
                table.append([g["lr"] for g in self.param_groups])
        self.regular_lrs_per_t = table
        if warmup_mode == "fix":
            assert isinstance(warmup_init_lr, float)
            start = [warmup_init_lr] * len(self.base_lrs)
        elif warmup_mode in ("factor", "auto"):
            assert isinstance(warmup_factor, float)
            start = [b * warmup_factor for b in self.base_lrs]
            if warmup_mode == "auto":
                self.warmup_end_lrs = table[-1]
        else:
            raise ValueError(f"Invalid warmup mode: {warmup_mode}")
        self._apply(start)

    @property
    def _is_plateau(self) -> bool:
        return isinstance(self.torch_scheduler, ReduceLROnPlateau)

    def _apply(self, lrs: List[float]) -> None:
        for g, lr in zip(self.param_groups, lrs):
            g["lr"] = lr

    def _ramp(self, t: int, regular: List[float]) -> List[float]:
        a = t / self.warmup_t
        if self.warmup_mode == "fix":
            return [self.warmup_init_lr * (1 - a) + b * a for b in self.base_lrs]
        if self.warmup_mode == "factor":
            f = self.warmup_factor * (1 - a) + a
            return [lr * f for lr in regular]
        return [b * self.warmup_factor * (1 - a) + e * a for b, e in zip(self.base_lrs, self.warmup_end_lrs)]

    def iter_update(self) -> None:
        """Call after every iteration."""
        if self.warmup_by_epoch:
            return
        self.last_iter += 1
        k = self.last_iter
        if k < self.warmup_t:
            self.in_iter_warmup = True
            self._apply(self._ramp(k, self.regular_lrs_per_t[k // self.epoch_len if self.by_epoch else k]))
        elif k == self.warmup_t:
            self._apply(self.regular_lrs_per_t[-1])
        else:
            self.in_iter_warmup = False
            if not self.by_epoch:
                self.torch_scheduler.step()

    def epoch_update(self, metric: Optional[float] = None) -> None:
        """Call after every epoch."""
        if not self.by_epoch:
            return
        self.last_epoch += 1
        e = self.last_epoch
        if self.warmup_by_epoch and e < self.warmup_t:
            self._apply(self._ramp(e, self.regular_lrs_per_t[e]))
        elif self.warmup_by_epoch and e == self.warmup_t:
            self._apply(self.regular_lrs_per_t[-1])
        elif not self.in_iter_warmup:
            self.torch_scheduler.step(metric) if self._is_plateau else self.torch_scheduler.step()

    def state_dict(self) -> Dict[str, Any]:
        st = {k: v for k, v in self.__dict__.items() if k != "torch_scheduler"}
        st["torch_scheduler"] = self.torch_scheduler.state_dict()
        return st

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        state_dict = dict(state_dict)
        self.torch_scheduler.load_state_dict(state_dict.pop("torch_scheduler"))
        self.__dict__.update(state_dict)
