"""Windowed metric store for the training loop and for checkpoints.

The reference's trainer pickles its metric store into every checkpoint and, on resume, installs whatever object the
file holds as the live store (reference cpu/trainer.py:406,469); its hooks then call ``update(iter, smooth, **values)``,
read ``values_maybe_smooth`` and index series by name for ``.avg`` / ``.latest`` / ``.global_avg`` / ``.global_sum``
(cpu/hooks/logger_hook.py:38-95, cpu/hooks/lr_update_hook.py:29-37).  A checkpoint written by this package must
therefore carry an object with exactly that protocol, or a reference-side resume dies at its first metric update.
This is that object; the implementation (a fixed ring per series, running totals) is this repo's.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple


class Series:
    """One scalar metric: last `window` values in a ring + running count / sum."""
    __slots__ = ("_ring", "_head", "_count", "_sum")

    def __init__(self, window: int = 20):
        self._ring = [0.0] * int(window)
        self._head = 0          # next write position
        self._count = 0
        self._sum = 0.0

    def update(self, value: float) -> None:
        v = float(value)
        self._ring[self._head] = v
        self._head = (self._head + 1) % len(self._ring)
        self._count += 1
        self._sum += v

    def _live(self):
        n = min(self._count, len(self._ring))
        return [self._ring[(self._head - 1 - i) % len(self._ring)] for i in range(n)]

    @property
    def latest(self) -> float:
        if not self._count:
            raise IndexError("empty metric series")
        return self._ring[(self._head - 1) % len(self._ring)]

    @property
    def avg(self) -> float:
        live = self._live()
        return sum(live) / len(live) if live else float("nan")

    @property
    def global_avg(self) -> float:
        return self._sum / self._count

    @property
    def global_sum(self) -> float:
        return self._sum

    def __getstate__(self):
        return {"ring": self._ring, "head": self._head, "count": self._count, "sum": self._sum}

    def __setstate__(self, st):
        self._ring, self._head, self._count, self._sum = list(st["ring"]), st["head"], st["count"], st["sum"]


class MetricStorage(dict):
    """name -> Series, with per-metric 'smooth' flag and last-iteration bookkeeping."""

    def __init__(self, window_size: int = 20):
        super().__init__()
        self._window = int(window_size)
        self._flags: Dict[str, bool] = {}
        self._last_iter: Dict[str, int] = {}

    def update(self, iter: Optional[int] = None, smooth: bool = True, **values) -> None:   # noqa: A002 (protocol name)
        for name, value in values.items():
            if name not in self._flags:
                self._flags[name] = bool(smooth)
                self._last_iter[name] = -1
                self[name] = Series(self._window)
            elif self._flags[name] != bool(smooth):
                raise AssertionError(f"metric '{name}' was registered with smooth={self._flags[name]}")
            if iter is None:
                self._last_iter[name] += 1
            else:
                if not iter > self._last_iter[name]:
                    raise AssertionError(f"metric '{name}': iteration {iter} is not after {self._last_iter[name]}")
                self._last_iter[name] = iter
            self[name].update(value)

    @property
    def values_maybe_smooth(self) -> Dict[str, Tuple[int, float]]:
        return {k: (self._last_iter[k], s.avg if self._flags[k] else s.latest) for k, s in self.items()}

    def __reduce__(self):      # dict subclass with extra attributes: make the pickle explicit and version-proof
        return (_rebuild, (self._window, dict(self), self._flags, self._last_iter))


def _rebuild(window, series, flags, last_iter):
    m = MetricStorage(window)
    dict.update(m, series)
    m._flags, m._last_iter = dict(flags), dict(last_iter)
    return m
