#!/usr/bin/env python3
"""Time tpsf_forward / tpsf_backward alone (HIP events, B samples):  python tools/tpsf_microbench.py [B] [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, c_int as I  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator().manual_seed(0)
depth = (torch.rand(B, 100, 100, generator=g) * 10).cuda()
ab = (torch.rand(B, 3, generator=g) * 0.5 + 0.7).cuda()
HR = torch.empty(B, 1, 100, 100, device="cuda")
LRd = torch.empty(B, 16, device="cuda")
psf = torch.empty(B, 1, 99, 99, device="cuda")
dl = torch.randn(B, 16, generator=g).cuda()
dab = torch.empty(B, 3, device="cuda")


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tf = timeit(lambda: call("tpsf_forward", ptr(depth), ptr(ab), ptr(HR), ptr(LRd), ptr(psf), I(B), stream()))
work = torch.empty(B * 10000, device="cuda")
tb = timeit(lambda: call("tpsf_backward", ptr(depth), ptr(ab), ptr(HR), ptr(dl), ptr(dab), ptr(work), I(B), stream()))
fb = 4 * (10000 + 10000 + 9801 + 16 + 3)
print(f"tpsf_forward  B={B}: {tf:.3f} ms  {B / tf / 1e3:.2f} M samples/s  {B * fb / tf / 1e6:.0f} GB/s ({B * fb / tf / 8e9 * 100:.1f}% of 8 TB/s)")
print(f"tpsf_backward B={B}: {tb:.3f} ms  {B / tb / 1e3:.2f} M samples/s  {B * 40000 / tb / 1e6:.0f} GB/s")
