#!/usr/bin/env python3
"""Time the tPSFNet MLP's GEMMs one by one (fp32 matrix cores, csrc/sgemm_mfma.hip) at batch B:
    python tools/sgemm_microbench.py [B]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, c_int as I, c_longlong as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator().manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = 0.0
for name, N, K in (("L1", 256, 48), ("L2", 1024, 256), ("L3", 256, 1024), ("L4", 3, 256)):
    x = torch.randn(B, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    b = torch.randn(N, generator=g).cuda()
    dy = torch.randn(B, N, generator=g).cuda()
    y = torch.empty(B, N, device="cuda")
    dx = torch.empty(B, K, device="cuda")
    ns = 32
    slab = torch.empty(ns * N * K, device="cuda")
    fl = 2.0 * B * N * K
    t_f = timed(lambda: call("tsr_sgemm", ptr(x), L(K), L(1), ptr(w), L(1), L(K), ptr(b), ptr(y), I(B), I(N), I(K), I(1), stream()))
    t_dx = timed(lambda: call("tsr_sgemm_masked", ptr(dy), L(N), L(1), ptr(w), L(K), L(1), ptr(x), ptr(dx), I(B), I(K), I(N), stream()))
    t_dw = timed(lambda: call("tsr_sgemm_splitk_strided", ptr(dy), L(1), L(N), ptr(x), L(K), L(1), ptr(slab), L(N * K), I(N), I(K),
                              I(B), I(ns), stream()))
    tot += t_f + t_dx + t_dw
    print(f"{name} {K}->{N} B={B}: fwd {t_f * 1e3:7.1f} us ({fl / t_f / 1e9:6.1f} TF)  dx {t_dx * 1e3:7.1f} us ({fl / t_dx / 1e9:6.1f} TF)  "
          f"dW {t_dw * 1e3:7.1f} us ({fl / t_dw / 1e9:6.1f} TF)")
print(f"sum {tot * 1e3:.1f} us")
