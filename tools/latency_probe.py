#!/usr/bin/env python3
"""Small-batch eval latency of TactileSR on the HIP path:  python tools/latency_probe.py [B ...]"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tactilesr_amd  # noqa: E402

torch.manual_seed(0)
m = tactilesr_amd.TactileSR().cuda().eval()
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 32, 128]:
    x = (torch.rand(B, 3, 4, 4) * 8).cuda()
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            y = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            y = m(x)
        e1.record()
        torch.cuda.synchronize()
    g = tactilesr_amd.GraphedTactileSR(m, B)
    for _ in range(5):
        g(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = g(x)
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / n
    print(f"B={B:4d}: eager {dt * 1e3:.3f} ms/forward ({B / dt:.0f} samples/s)   HIP graph {dg * 1e3:.3f} ms ({B / dg:.0f} samples/s)")
