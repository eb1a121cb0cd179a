"""torch.optim.Adam-compatible optimizer whose step runs the fused HIP kernel
``tsr_adam_l2_step`` (L2-in-gradient weight decay, i.e. torch.optim.Adam, not AdamW --
reference train/tactileSR_train.py:212: ``optim.Adam(model.parameters(), lr, weight_decay)``).

It subclasses ``torch.optim.Optimizer`` so ``param_groups`` / ``state_dict`` / lr schedulers
(``optim.lr_scheduler.StepLR``, the reference's LRWarmupScheduler) work unchanged; the state keys
(``step``, ``exp_avg``, ``exp_avg_sq``) are torch.optim.Adam's, so checkpoints interoperate
(cpu/trainer.py:401-421).
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream, c_int as _I, c_float as _F, c_longlong as _L, TactileSRHipError


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise TactileSRHipError("tactilesr_amd.optim.Adam needs fp32 parameters on a ROCm device")
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                call("tsr_adam_l2_step", ptr(p.data), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]),
                     _L(p.numel()), _F(group["lr"]), _F(b1), _F(b2), _F(group["eps"]), _F(group["weight_decay"]),
                     _I(int(st["step"].item())), stream())
                # the kernel wrote p behind autograd's back: bump the version counter so cached
                # weight packs (TactileSR._plan) are rebuilt, without launching anything
                torch._C._autograd._unsafe_set_version_counter((p,), (p._version + 1,))
        return loss
