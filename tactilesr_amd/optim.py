"""torch.optim.Adam-compatible optimizer whose step is ONE launch of the fused HIP kernel ``tsr_adam_l2_multi``
over all parameter tensors (L2-in-gradient weight decay, i.e. torch.optim.Adam, not AdamW -- reference
train/tactileSR_train.py:212: ``optim.Adam(model.parameters(), lr, weight_decay)``; stepped at cpu/trainer.py:361).

It subclasses ``torch.optim.Optimizer`` so ``param_groups`` / ``state_dict`` / lr schedulers
(``optim.lr_scheduler.StepLR``, the warm-up wrapper) work unchanged; the state keys (``step``, ``exp_avg``,
``exp_avg_sq``) are torch.optim.Adam's, so checkpoints interoperate (cpu/trainer.py:401-421).

Per step the host builds nothing: a device table of (param, grad, exp_avg, exp_avg_sq, n) records, one per <= 4096
elements, is cached per set of live (param, grad) addresses -- with the engine's gradient arena those are the same
every step -- and the kernel walks it.  Parameters without a gradient are skipped exactly like torch does (the
Seqs transplant leaves an optimizer holding discarded modules, train/tactileSRSeqs_train.py:74-77).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream, c_int as _I, c_float as _F, TactileSRHipError

CHUNK = 4096


class _Rec(ctypes.Structure):         # mirror of tsr_adam_chunk (include/tactilesr_hip.h)
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int), ("reserved", ctypes.c_int)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}           # key (every address the table holds) -> (device table, n_chunks)
        self.launches = 0           # kernel launches issued so far (tests: one per step)
        self.table_builds = 0       # device tables built so far (steady state with the gradient arena: exactly one)

    def _table(self, items):
        key = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    for p, g, st in items)
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        recs = []
        for p, g, st in items:
            n = p.numel()
            for off in range(0, n, CHUNK):
                b = 4 * off
                recs.append((p.data_ptr() + b, g.data_ptr() + b, st["exp_avg"].data_ptr() + b,
                             st["exp_avg_sq"].data_ptr() + b, min(CHUNK, n - off), 0))
        arr = (_Rec * len(recs))(*recs)
        host = torch.frombuffer(memoryview(arr).cast("B"), dtype=torch.uint8).clone()
        dev = host.to(items[0][0].device)
        if len(self._tables) > 8:
            self._tables.clear()
        # The key IS the table: every address a record holds (param, grad, exp_avg, exp_avg_sq of every tensor) is part
        # of it, so a hit is valid whatever lived at those addresses in between -- no tensor has to be kept alive for
        # it (holding the gradients here pinned up to nine stale gradient sets of a model whose gradients are fresh
        # autograd tensors every step).
        self._tables[key] = (dev, len(recs))
        self.table_builds += 1
        return self._tables[key]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise TactileSRHipError("tactilesr_amd.optim.Adam needs fp32 parameters on a ROCm device")
                if not p.is_contiguous():
                    raise TactileSRHipError("tactilesr_amd.optim.Adam needs contiguous parameters")
                g = p.grad
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append((p, g, st))     # CPU scalar: no device sync
            for step, items in by_step.items():
                dev, n_chunks = self._table(items)
                call("tsr_adam_l2_multi", ptr(dev), _I(n_chunks), _F(group["lr"]), ctypes.c_double(b1), ctypes.c_double(b2),
                     _F(group["eps"]),
                     _F(group["weight_decay"]), _I(step), stream())
                self.launches += 1
        # the kernel wrote the parameters behind autograd's back: invalidate cached weight packs (TactileSR._plan)
        _lib.bump_param_epoch()
        return loss
