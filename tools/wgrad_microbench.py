#!/usr/bin/env python3
"""Time tsr_conv2d_wgrad_bf16s alone, on the operand pattern the train step gives it (input = relu(bn(z)) fused into
the staging, gradient-like dz), and sanity-check the result against torch's conv weight gradient on the GPU:

    python tools/wgrad_microbench.py [ks cin cout B planes [splits]]        (WG_RAW=1: no fused input transform)
"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, load, c_int as I, c_float as Fl, c_longlong as L  # noqa: E402
from tactilesr_amd.model.tactileSR_model import to_cb16  # noqa: E402

ks, cin, cout, B, planes = [int(x) for x in (sys.argv[1:6] + ["5", "128", "128", "1024", "-2"][len(sys.argv) - 1:])]
H = W = 40
g = torch.Generator().manual_seed(0)
z = torch.randn(B, cin, H, W, generator=g).cuda()
dzn = (torch.randn(B, cout, H, W, generator=g) * 1e-3).cuda()
sc = (torch.rand(cin, generator=g) + 0.5).cuda()
sh = (torch.randn(cin, generator=g) * 0.3).cuda()
a, dz = to_cb16(z), to_cb16(dzn)
am = torch.stack([z.abs().max(), dzn.abs().max()])
ns = load().tsr_conv2d_wgrad_splits(cout, cin, ks, planes, B, H, W)
if len(sys.argv) > 6:
    ns = int(sys.argv[6])
n = cout * cin * ks * ks
slab = torch.empty(ns * n, device="cuda")
bslab = torch.empty(ns * cout, device="cuda")
if planes == -1:
    a, dz = a.to(torch.bfloat16), dz.to(torch.bfloat16)


RAW = os.environ.get("WG_RAW") == "1"
if RAW:
    sc = sh = None


def run():
    call("tsr_conv2d_wgrad_bf16s", ptr(a), I(cin), I(0), I(cin), ptr(sc), ptr(sh), ptr(dz), I(cout), I(0), I(cout),
         I(ks), I(planes), ptr(am[0:1]), ptr(am[1:2]), ptr(slab), ptr(bslab), I(ns), I(B), I(H), I(W), stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 10
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
out = torch.empty(cout, cin, ks, ks, device="cuda")
call("tsr_reduce_splits", ptr(slab), ptr(out), L(n), I(ns), Fl(1.0), stream())
nb = min(B, 64)                                     # sanity reference on the first images only when B is large
if nb == B:
    act = (z if RAW else F.relu(z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))).double()
    ref = torch.nn.grad.conv2d_weight(act, (cout, cin, ks, ks), dzn.double(), padding=ks // 2)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
else:
    err = float("nan")
fl = 2.0 * B * H * W * cin * cout * ks * ks
print(f"wgrad {ks}x{ks} {cin}->{cout} B={B} planes={planes} splits={ns}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF algorithmic "
      f"({fl * (3 if planes == -2 else 6 if planes == 3 else 1) / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF executed)  "
      f"max-norm err vs torch fp64 {err:.1e}")
