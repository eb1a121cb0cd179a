"""MI355X-native drop-in for the reference's ``model/tPSFNet.py``.

``tPSFNet(gama, perception_scale, size=(100,100), device=None)`` keeps the reference's
constructor, attributes (``gama``, ``perception_scale``, ``MLP_layer``, ``PSF_sdf``,
``LR_masking_sdf``), ``state_dict`` keys (``MLP_layer.{1,3,5,7}.{weight,bias}``), init
(Linear weights N(0, 0.03), default biases; reference :57-65) and the 4-tuple return of
``forward(x, depth)`` -> ``(HR, LR_deg, psf, alphaBeta)`` (:102-127).  The python loop over the
batch is replaced by one batched HIP launch (the separable PSF convolution as two Toeplitz GEMMs on the
matrix cores, csrc/tpsf_mfma.hip), the MLP runs on an fp32-MFMA SGEMM, and the
whole forward is one ``autograd.Function`` so ``Trainer_tPSF.train_cal_loss``'s
``MSE(LR[:,2:3], LR_deg).backward()`` (train/tPSFNet_train.py:180-190) reaches the MLP weights.
No CPU fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _lib
from .._lib import call, ptr, stream, c_int as _I, c_longlong as _L


def _linear(x, w, b, act):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    call("tsr_sgemm", ptr(x), _L(K), _L(1), ptr(w), _L(1), _L(K), ptr(b), ptr(y), _I(M), _I(N), _I(K), _I(act), stream())
    return y


_ALIGN = 64      # gradient slots start on 256-B boundaries (the fused Adam kernel's 16-B accesses)


class _GradPlan:
    """Where the MLP's parameter gradients land: ONE flat fp32 buffer holding the eight tensors in parameter order (slot
    offsets 256-B aligned) and ONE split-K slab `[nsplit][total]` that every dW / db launch of a backward writes its
    partial sums into, so that a single ``tsr_reduce_splits`` adds all of them in a fixed order.  ``p.grad`` aliases the
    flat buffer (autograd adopts the fresh views it is handed), so the addresses the fused Adam table holds never change:
    the table is built once (gradients that are fresh autograd tensors every step re-built it per step)."""

    def __init__(self, params, device):
        self.shapes = [tuple(p.shape) for p in params]
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.slab = None
        self.nsplit = 0

    def matches(self, params, device):
        return self.flat.device == device and self.shapes == [tuple(p.shape) for p in params]

    def slab_for(self, nsplit):
        if self.slab is None or self.nsplit != nsplit:
            self.slab = torch.zeros(nsplit * self.total, dtype=torch.float32, device=self.flat.device)
            self.nsplit = nsplit
        return self.slab

    def view(self, i, buf=None):
        buf = self.flat if buf is None else buf
        n = 1
        for d in self.shapes[i]:
            n *= d
        return buf[self.offsets[i]:self.offsets[i] + n].view(self.shapes[i])


class _Token:
    """Identity of one differentiable forward (weakly referenced by ddp.note_forward)."""


class _TPSFFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, depth, *params):
        ws, bs = params[0::2], params[1::2]
        B = x.shape[0]
        h = [x.reshape(B, -1).float().contiguous()]
        for i in range(4):
            h.append(_linear(h[-1], ws[i].detach().contiguous(), bs[i].detach().contiguous(), 1 if i < 3 else 2))
        ab = h[-1]
        d = depth.detach().float().contiguous()
        HR = torch.empty(B, 1, 100, 100, dtype=torch.float32, device=x.device)
        LRd = torch.empty(B, 1, 4, 4, dtype=torch.float32, device=x.device)
        psf = torch.empty(B, 1, 99, 99, dtype=torch.float32, device=x.device)
        call("tpsf_forward", ptr(d), ptr(ab), ptr(HR), ptr(LRd), ptr(psf), _I(B), stream())
        ctx.h, ctx.d, ctx.params, ctx.owner = h, d, params, owner
        ctx.token = _Token()
        if any(ctx.needs_input_grad):
            from ..ddp import note_forward
            note_forward(owner, ctx.token)     # the flat gradient buffer goes only to the sole pending backward
        ctx.save_for_backward(HR)          # the backward's reductions read the stored output: an in-place edit of the
        ctx.mark_non_differentiable(HR, psf)      # returned HR before backward() trips autograd's version check
        ctx.set_materialize_grads(False)   # no zero-filled (B,1,100,100) / (B,1,99,99) gradients for the unused outputs
        return HR, LRd, psf, ab.view(B, 1, 3).clone()

    @staticmethod
    def backward(ctx, gHR, gLR, gpsf, gab):
        h, d, params, owner = ctx.h, ctx.d, ctx.params, ctx.owner
        ws = params[0::2]
        B = h[0].shape[0]
        dev = d.device
        dab = torch.empty(B, 3, dtype=torch.float32, device=dev)
        if gLR is not None:
            work = torch.empty(B * 10000, dtype=torch.float32, device=dev)
            call("tpsf_backward", ptr(d), ptr(h[-1]), ptr(ctx.saved_tensors[0]), ptr(gLR.contiguous().float()), ptr(dab),
                 ptr(work), _I(B), stream())
        else:
            dab.zero_()
        if gab is not None:
            dab = dab + gab.reshape(B, 3)
        # gradient plan: the flat buffer is handed to autograd only when no .grad exists yet (zero_grad()'s default) and
        # no other backward of this module is pending; otherwise (accumulation, a module applied twice in one graph)
        # fresh tensors, which autograd adds up itself
        plan = owner._grad_plan
        if plan is None or not plan.matches(params, dev):
            plan = owner._grad_plan = _GradPlan(params, dev)
        live = getattr(owner, "_live_forwards", None)
        alone = True
        if live is not None:
            alone = not any(t is not ctx.token for t in live)
            live.discard(ctx.token)
        shared = owner._shared_graph or not alone
        owner._shared_graph = shared and not alone
        direct = not shared and all(p.grad is None for p in params)
        out = plan.flat if direct else torch.empty_like(plan.flat)
        ns = max(1, min(32, (B + 15) // 16))
        slab = plan.slab_for(ns)
        tot = plan.total
        dy = dab.contiguous()
        call("tsr_act_bwd", ptr(dy), ptr(h[4]), _L(dy.numel()), _I(2), stream())           # Softplus'
        for i in reversed(range(4)):
            w = ws[i].detach().contiguous()
            N, K = w.shape
            # dW = dy^T x and db = 1^T dy (reductions over the batch): partial sums per batch range into the shared slab
            call("tsr_sgemm_splitk_strided", ptr(dy), _L(1), _L(N), ptr(h[i]), _L(K), _L(1),
                 ptr(slab[plan.offsets[2 * i]:]), _L(tot), _I(N), _I(K), _I(B), _I(ns), stream())
            call("tsr_colsum_splitk", ptr(dy), ptr(slab[plan.offsets[2 * i + 1]:]), _L(tot), _I(B), _I(N), _I(ns), stream())
            if i > 0:
                # dx = (dy W) masked by the ReLU of the layer below (its stored output h[i]): one launch
                dx = torch.empty(B, K, dtype=torch.float32, device=dev)
                call("tsr_sgemm_masked", ptr(dy), _L(N), _L(1), ptr(w), _L(K), _L(1), ptr(h[i]), ptr(dx), _I(B), _I(K),
                     _I(N), stream())
                dy = dx
        call("tsr_reduce_splits", ptr(slab), ptr(out), _L(tot), _I(ns), _lib.c_float(1.0), stream())
        return (None, None, None) + tuple(plan.view(i, out) for i in range(8))


class tPSFNet(nn.Module):
    def __init__(self, gama, perception_scale, size=(100, 100), device=None):
        super().__init__()
        assert tuple(size) == (100, 100), "tPSFNet geometry is fixed at 100x100 (reference model/tPSFNet.py:40-55)"
        self.gama = gama
        self.perception_scale = perception_scale
        self.device = device
        self.MLP_layer = nn.Sequential(
            nn.Flatten(), nn.Linear(16 * 3, 256), nn.ReLU(), nn.Linear(256, 1024), nn.ReLU(),
            nn.Linear(1024, 256), nn.ReLU(), nn.Linear(256, 3), nn.Softplus())
        for m in self.MLP_layer:
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.03)
        self.zeroPad_func = nn.ZeroPad2d(padding=(48, 48, 48, 48))
        # geometry constants kept as plain attributes like the reference (not buffers: not in state_dict)
        u = torch.arange(99, dtype=torch.float32)
        sdf = ((u.view(-1, 1) - 49) ** 2 + (u.view(1, -1) - 49) ** 2) ** 0.5
        self.PSF_sdf = (10 * (sdf - sdf.min()) / (sdf.max() - sdf.min())).view(1, 1, 99, 99)
        xs = torch.arange(100, dtype=torch.float32)
        m = torch.zeros(4, 4, 100, 100)
        for a in range(4):
            for b in range(4):
                m[a, b] = ((xs.view(-1, 1) - (12 + 25 * a)) ** 2 + (xs.view(1, -1) - (12 + 25 * b)) ** 2) ** 0.5
        self.LR_masking_sdf = 10 * (m - m.min()) / (m.max() - m.min())
        self._grad_plan = None            # flat gradient buffer + split-K slab (created by the first backward)
        self._shared_graph = False
        assert abs(float(sdf.max()) - math.sqrt(4802.0)) < 1e-3 and abs(float(m.max()) - math.sqrt(15138.0)) < 1e-3

    def forward(self, x, depth):
        assert x.shape[0] == depth.shape[0], "Batch size of LR tactile and depth should be the same!"
        if not (x.is_cuda and depth.is_cuda):
            raise _lib.TactileSRHipError("tPSFNet (tactilesr_amd) runs on MI355X only (no CPU fallback)")
        lin = [self.MLP_layer[i] for i in (1, 3, 5, 7)]
        params = []
        for l in lin:
            params += [l.weight, l.bias]
        return _TPSFFn.apply(self, x, depth, *params)
