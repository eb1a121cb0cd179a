"""LR warm-up in front of a torch scheduler, as a PRECOMPUTED TABLE.

Role: what the reference's trainer builds from the ``warmup_*`` arguments of train/tactileSR_train.py:215-228
(its vendored ``cpu/lr_scheduler.py``).  Only the constructor signature, ``iter_update`` / ``epoch_update`` and the
learning-rate sequence are the contract (``tests/golden/lr_schedule.npz`` holds the reference's own sequences for
five configurations; this class reproduces them bit for bit).  The design is this repo's:

* one *warm-up clock* -- iterations or epochs -- and a table ``warm[k]`` (k = 1..warm_ticks) holding the complete
  learning-rate vector of every warm-up tick, computed once in the constructor.  A warm-up tick is a table lookup;
* after the table is exhausted, ticks of the scheduler's own clock are forwarded to the wrapped torch scheduler;
* the state is three integers plus the wrapped scheduler's state (``state_dict``); a state written by the
  reference's class is accepted too (``load_state_dict`` recognises its ``last_iter`` / ``last_epoch`` keys), and
  ``reference_state_dict`` writes that layout for a reference-side resume.

Observable quirks of the sequence, all covered by the fixture: building the table has to ADVANCE the wrapped
scheduler (a torch scheduler can only be queried by stepping it), so after warm-up it continues from the epoch the
warm-up ended in; "auto" ignores ``warmup_init_lr`` and ramps from ``base_lr * warmup_factor`` to the wrapped
scheduler's rate at the end of warm-up; with an epoch scheduler under an iteration warm-up, epoch ends inside the
warm-up window do not step the wrapped scheduler (the table already accounts for them).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

from torch.optim.lr_scheduler import ReduceLROnPlateau

_MODES = ("fix", "auto", "factor")


def _lerp(lo: float, hi: float, a: float) -> float:
    # expression order is part of the bit-exact contract with the fixture
    return lo * (1 - a) + hi * a


def warmup_table(base: Sequence[float], plain: List[List[float]], ticks: int, ticks_per_plain: int, mode: str,
                 init_lr: Optional[float], factor: Optional[float]) -> List[List[float]]:
    """Learning-rate vectors for warm-up ticks 0..ticks.

    ``plain[j]`` is the wrapped scheduler's vector after j of its own steps; tick k sits in plain step
    ``k // ticks_per_plain``.  Entry 0 is the start value, entry ``ticks`` is ``plain[-1]`` (warm-up over)."""
    if mode == "fix":
        if not isinstance(init_lr, float):
            raise AssertionError("warmup_mode='fix' needs a float warmup_init_lr")
        start = [init_lr for _ in base]
    elif mode in ("factor", "auto"):
        if not isinstance(factor, float):
            raise AssertionError(f"warmup_mode='{mode}' needs a float warmup_factor")
        start = [b * factor for b in base]
    else:
        raise ValueError(f"Invalid warmup mode: {mode}")
    rows = [start]
    end = plain[-1]
    for k in range(1, ticks):
        a = k / ticks
        if mode == "fix":
            rows.append([_lerp(init_lr, b, a) for b in base])
        elif mode == "factor":
            f = _lerp(factor, 1.0, a)
            rows.append([lr * f for lr in plain[k // ticks_per_plain]])
        else:
            rows.append([_lerp(b * factor, e, a) for b, e in zip(base, end)])
    rows.append(list(end))
    return rows


class LRWarmupScheduler:
    def __init__(self, torch_scheduler, by_epoch: bool = True, epoch_len: Optional[int] = None,
                 warmup_t: int = 0, warmup_by_epoch: bool = False, warmup_mode: str = "fix",
                 warmup_init_lr: Optional[float] = None, warmup_factor: Optional[float] = None):
        self.torch_scheduler = torch_scheduler
        self.by_epoch = bool(by_epoch)
        self.epoch_len = epoch_len
        self._plateau = isinstance(torch_scheduler, ReduceLROnPlateau)
        if warmup_by_epoch and not by_epoch:
            raise AssertionError("an epoch warm-up needs an epoch-based scheduler")
        if self._plateau and not by_epoch:
            raise AssertionError("ReduceLROnPlateau is epoch-based")
        base = [g["lr"] for g in self._groups]
        self._ticks = int(warmup_t or 0)                       # length of the warm-up on its clock
        self._clock = None if not self._ticks else ("epoch" if warmup_by_epoch else "iter")
        self._n_iter = 0
        self._n_epoch = 0
        self._warm: List[List[float]] = []
        if self._clock is None:
            return
        per_plain = 1
        if self._clock == "iter" and self.by_epoch:
            if epoch_len is None:
                raise AssertionError("an iteration warm-up in front of an epoch scheduler needs epoch_len")
            per_plain = int(epoch_len)
        plain = [list(base)]
        for _ in range(self._ticks // per_plain):
            if not self._plateau:
                torch_scheduler.step()
            plain.append([g["lr"] for g in self._groups] if not self._plateau else list(base))
        self._warm = warmup_table(base, plain, self._ticks, per_plain, warmup_mode, warmup_init_lr, warmup_factor)
        self._set(self._warm[0])

    # ------------------------------------------------------------------ helpers
    @property
    def _is_plateau(self) -> bool:
        """Read by the reference's LRUpdateHook.after_epoch on every epoch (cpu/hooks/lr_update_hook.py:33;
        the reference class exposes it as a property too, cpu/lr_scheduler.py:93-95)."""
        return self._plateau

    @property
    def _groups(self):
        # looked up on every use: optimizer.load_state_dict() replaces the param_group dicts
        return self.torch_scheduler.optimizer.param_groups

    def _set(self, lrs: Sequence[float]) -> None:
        for g, lr in zip(self._groups, lrs):
            g["lr"] = lr

    def _step_wrapped(self, metric=None) -> None:
        if self._plateau:
            self.torch_scheduler.step(metric)
        else:
            self.torch_scheduler.step()

    @property
    def in_warmup(self) -> bool:
        n = self._n_iter if self._clock == "iter" else self._n_epoch
        return self._clock is not None and n < self._ticks

    # ------------------------------------------------------------------ the two clocks
    def iter_update(self) -> None:
        """Call after every iteration."""
        if self._clock == "epoch":
            return
        self._n_iter += 1
        if self._n_iter <= self._ticks:
            self._set(self._warm[self._n_iter])
        elif not self.by_epoch:
            self._step_wrapped()

    def epoch_update(self, metric: Optional[float] = None) -> None:
        """Call after every epoch."""
        if not self.by_epoch:
            return
        self._n_epoch += 1
        if self._clock == "epoch":
            if self._n_epoch <= self._ticks:
                self._set(self._warm[self._n_epoch])
            else:
                self._step_wrapped(metric)
            return
        # iteration warm-up: the table covers every epoch end up to and including the tick that ends it
        covered = self._clock == "iter" and self._ticks > 1 and 0 < self._n_iter <= self._ticks
        if not covered:
            self._step_wrapped(metric)

    # ------------------------------------------------------------------ state
    def state_dict(self) -> Dict[str, Any]:
        return {"n_iter": self._n_iter, "n_epoch": self._n_epoch, "warm_ticks": self._ticks,
                "torch_scheduler": self.torch_scheduler.state_dict()}

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        state = dict(state)
        self.torch_scheduler.load_state_dict(state.pop("torch_scheduler"))
        if "last_iter" in state or "last_epoch" in state:        # written by the reference's class
            self._n_iter, self._n_epoch = int(state.get("last_iter", 0)), int(state.get("last_epoch", 0))
            return
        if int(state["warm_ticks"]) != self._ticks:
            raise ValueError("checkpoint was written with a different warm-up length")
        self._n_iter, self._n_epoch = int(state["n_iter"]), int(state["n_epoch"])

    def reference_state_dict(self) -> Dict[str, Any]:
        """The counters in the key layout the reference's class restores with ``__dict__.update`` (only the keys
        its ``iter_update`` / ``epoch_update`` read after a resume; its constructor rebuilds the rest)."""
        held = self._clock == "iter" and self._ticks > 1 and 0 < self._n_iter <= self._ticks
        return {"last_iter": self._n_iter, "last_epoch": self._n_epoch, "in_iter_warmup": bool(held),
                "torch_scheduler": self.torch_scheduler.state_dict()}
