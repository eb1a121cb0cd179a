"""HIP-backed functional pieces of the SR training / eval step that sit outside the model:
target preparation, MSE loss, PSNR / SSIM (reference train/tactileSR_train.py:41-51,66-101;
utility/tools.py:49-81)."""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream, c_int as _I, c_float as _F, c_longlong as _L, c_double as _D, TactileSRHipError


def prepare_target(HR_raw: torch.Tensor, HR_scale_num: float = 10.0, scale_factor: int = 10) -> torch.Tensor:
    """``HR.float()/HR_scale_num`` + ``F.interpolate(HR, (4sf,4sf), bilinear)`` in one kernel
    (train/tactileSR_train.py:44-45)."""
    if not HR_raw.is_cuda:
        raise TactileSRHipError("prepare_target needs a ROCm tensor (no CPU fallback)")
    hr = HR_raw.detach().float().contiguous()
    B, C, hin, win = hr.shape
    assert C == 1
    H = W = 4 * scale_factor
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=hr.device)
    call("tsr_target_prep", ptr(hr), ptr(out), _F(1.0 / HR_scale_num), _I(B), _I(hin), _I(win), _I(H), _I(W), stream())
    return out


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, target):
        y = y.contiguous()
        t = target.detach().float().contiguous()
        dy = torch.empty_like(y)
        loss = torch.empty(1, dtype=torch.float32, device=y.device)
        work = torch.empty(256, dtype=torch.float64, device=y.device)
        call("tsr_mse_fwd_bwd", ptr(y), ptr(t), ptr(dy), ptr(loss), _L(y.numel()), _F(1.0), ptr(work), stream())
        ctx.save_for_backward(dy)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (dy,) = ctx.saved_tensors
        return dy * gout, None


def mse_loss(y: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """``nn.MSELoss()(y, target)`` (mean over all elements): loss and d loss/d y from one pass."""
    if not y.is_cuda:
        raise TactileSRHipError("mse_loss needs ROCm tensors (no CPU fallback)")
    return _MSE.apply(y.float(), target)


def psnr_ssim(a: torch.Tensor, b: torch.Tensor, maxValue: float, reference_quirk: bool = True,
              C1: float = 0.01 ** 2, C2: float = 0.03 ** 2):
    """Per-sample ``calculationPSNR`` / ``calculationSSIM`` for (B,1,H,W) batches.  With
    ``reference_quirk`` the squared error is divided by shape[0]*shape[1] of the (1,H,W) slice the
    reference's eval_func passes (= H, not H*W; utility/tools.py:60-61, train/tactileSR_train.py:89)."""
    a = a.detach().float().contiguous()
    b = b.detach().float().contiguous()
    B, H, W = a.shape[0], a.shape[-2], a.shape[-1]
    n = a.numel() // B
    div = float(1 * H) if reference_quirk else float(H * W)
    ps = torch.empty(B, dtype=torch.float32, device=a.device)
    ss = torch.empty(B, dtype=torch.float32, device=a.device)
    call("tsr_psnr_ssim", ptr(a), ptr(b), _I(B), _I(n), _D(div), _D(maxValue), _D(C1), _D(C2), ptr(ps), ptr(ss), stream())
    return ps, ss
