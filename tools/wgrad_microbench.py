#!/usr/bin/env python3
"""Time tsr_conv2d_wgrad_bf16s alone:  python tools/wgrad_microbench.py [ks cin cout B planes]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, c_int as I  # noqa: E402

ks, cin, cout, B, planes = [int(x) for x in (sys.argv[1:6] + ["5", "128", "128", "1024", "-2"][len(sys.argv) - 1:])]
H = W = 40
g = torch.Generator().manual_seed(0)
a = torch.randn(B * cin * H * W, generator=g).cuda()
dz = (torch.randn(B * cout * H * W, generator=g) * 1e-3).cuda()
am = torch.tensor([6.0, 6e-3]).cuda()
from tactilesr_amd._lib import load  # noqa: E402
if os.environ.get("TSR_WGRAD_OLD"):
    slices = ks * (cout // 64) * (cin // 64)
    ns = max(1, min(B * 25, 1024 // slices))
else:
    ns = load().tsr_conv2d_wgrad_splits(cout, cin, ks, planes, B, H, W)
if len(sys.argv) > 6:
    ns = int(sys.argv[6])
slab = torch.empty(ns * cout * cin * ks * ks, device="cuda")
bslab = torch.empty(ns * cout, device="cuda")


def run():
    call("tsr_conv2d_wgrad_bf16s", ptr(a), I(cin), I(0), I(cin), ptr(None), ptr(None), ptr(dz), I(cout), I(0), I(cout),
         I(ks), I(planes), ptr(am[0:1]), ptr(am[1:2]), ptr(slab), ptr(bslab), I(ns), I(B), I(H), I(W), stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * H * W * cin * cout * ks * ks
print(f"wgrad {ks}x{ks} {cin}->{cout} B={B} planes={planes} splits={ns}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF algorithmic "
      f"({fl * (3 if planes == -2 else 6 if planes == 3 else 1) / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF executed)")
