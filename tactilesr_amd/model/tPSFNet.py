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


def _splitk(a, sa0, sa1, b, sb0, sb1, out, M, N, K):
    """out[M][N] = sum_k a(i,k) b(k,j) with K split over enough workgroups to fill the chip; the partial slabs
    are added in a fixed order (tsr_reduce_splits)."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    ns = max(1, min((1024 + tiles - 1) // tiles, (K + 63) // 64))
    slab = torch.empty(ns * M * N, dtype=torch.float32, device=out.device)
    call("tsr_sgemm_splitk", ptr(a), _L(sa0), _L(sa1), ptr(b), _L(sb0), _L(sb1), ptr(slab), _I(M), _I(N), _I(K), _I(ns),
         stream())
    call("tsr_reduce_splits", ptr(slab), ptr(out), _L(M * N), _I(ns), _lib.c_float(1.0), stream())


class _TPSFFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, depth, *params):
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
        ctx.h, ctx.d, ctx.params = h, d, params
        ctx.save_for_backward(HR)          # the backward's reductions read the stored output: an in-place edit of the
        ctx.mark_non_differentiable(HR, psf)      # returned HR before backward() trips autograd's version check
        return HR, LRd, psf, ab.view(B, 1, 3).clone()

    @staticmethod
    def backward(ctx, gHR, gLR, gpsf, gab):
        h, d, params = ctx.h, ctx.d, ctx.params
        ws = params[0::2]
        B = h[0].shape[0]
        dab = torch.zeros(B, 3, dtype=torch.float32, device=d.device)
        if gLR is not None:
            work = torch.empty(B * 10000, dtype=torch.float32, device=d.device)
            call("tpsf_backward", ptr(d), ptr(h[-1]), ptr(ctx.saved_tensors[0]), ptr(gLR.contiguous().float()), ptr(dab), ptr(work),
                 _I(B), stream())
        if gab is not None:
            dab = dab + gab.reshape(B, 3)
        grads = []
        dy = dab.contiguous()
        ones = torch.ones(B, 1, dtype=torch.float32, device=d.device)
        for i in reversed(range(4)):
            call("tsr_act_bwd", ptr(dy), ptr(h[i + 1]), _L(dy.numel()), _I(1 if i < 3 else 2), stream())
            w = ws[i].detach().contiguous()
            N, K = w.shape
            gw = torch.empty(N, K, dtype=torch.float32, device=d.device)      # dW = dy^T x   (reduction over the batch)
            _splitk(dy, 1, N, h[i], K, 1, gw, N, K, B)
            gb = torch.empty(1, N, dtype=torch.float32, device=d.device)      # db = 1^T dy
            _splitk(ones, 1, 1, dy, N, 1, gb, 1, N, B)
            grads = [gw, gb.view(N)] + grads
            if i > 0:
                dx = torch.empty(B, K, dtype=torch.float32, device=d.device)  # dx = dy W
                call("tsr_sgemm", ptr(dy), _L(N), _L(1), ptr(w), _L(K), _L(1), ptr(None), ptr(dx), _I(B), _I(K), _I(N),
                     _I(0), stream())
                dy = dx
        return (None, None) + tuple(grads)


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
        assert abs(float(sdf.max()) - math.sqrt(4802.0)) < 1e-3 and abs(float(m.max()) - math.sqrt(15138.0)) < 1e-3

    def forward(self, x, depth):
        assert x.shape[0] == depth.shape[0], "Batch size of LR tactile and depth should be the same!"
        if not (x.is_cuda and depth.is_cuda):
            raise _lib.TactileSRHipError("tPSFNet (tactilesr_amd) runs on MI355X only (no CPU fallback)")
        lin = [self.MLP_layer[i] for i in (1, 3, 5, 7)]
        params = []
        for l in lin:
            params += [l.weight, l.bias]
        return _TPSFFn.apply(x, depth, *params)
