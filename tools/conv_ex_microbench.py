#!/usr/bin/env python3
"""Time one TRAINING-path convolution launch (tsr_conv2d_ex, fp16x3):  python tools/conv_ex_microbench.py [ks cin cout B]

  fwd   : epi_mode 1 (raw output + BatchNorm statistics), input = relu(z*scale+shift) applied while staging
  dgrad : epi_mode 2 (ReLU mask by the stored activation + BatchNorm-backward sums), plain input
  plain : epi_mode 0 through the EXT instantiation, plain input
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd import _lib  # noqa: E402
from tactilesr_amd.model._train import conv_ex, _pack, Act  # noqa: E402

argv = sys.argv[1:]
ks, cin, cout, B = [int(x) for x in (argv[:4] + ["5", "128", "128", "2048"][len(argv[:4]):])]
H = W = 40
g = torch.Generator().manual_seed(0)
x = torch.randn(B * cin * H * W, generator=g).cuda()
w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
wamax = w.abs().max().reshape(1)
wp = _pack(w, cout, cin, ks, -2, wamax)
sc = (torch.rand(cin, generator=g) + 0.5).cuda()
sh = (torch.randn(cin, generator=g) * 0.3).cuda()
amax = torch.stack([x.abs().max(), torch.zeros((), device="cuda")]).contiguous()
out = torch.empty(B * cout * H * W, device="cuda")
lib = _lib.load()
entries = lib.tsr_conv2d_slab_entries_ex(B, H, W, cout, ks, -2)
slab = torch.empty(entries * cout * 2, device="cuda")
cnt = torch.empty(entries, device="cuda")
z = torch.randn(B * cout * H * W, generator=g).cuda()
msc = (torch.rand(cout, generator=g) + 0.5).cuda()
msh = (torch.randn(cout, generator=g) * 0.3).cuda()
xa, xb = torch.rand(cout, device="cuda"), torch.rand(cout, device="cuda")


def fwd():
    conv_ex(B=B, H=H, W=W, src=Act(x, cin, 0, cin, sc, sh, amax=amax[0:1]), w=wp, cout=cout, ks=ks, out=out,
            out_ctot=cout, out_coff=0, epi_mode=1, slab=slab, slab_cnt=cnt, nsplit=-2, out_amax=amax[1:2], w_amax=wamax)


def dgrad():
    conv_ex(B=B, H=H, W=W, src=Act(x, cin, 0, cin, amax=amax[0:1]), w=wp, cout=cout, ks=ks, out=out, out_ctot=cout,
            out_coff=0, epi_mode=2, mask=Act(z, cout, 0, cout, msc, msh, xa, xb), bn=True, slab=slab, slab_cnt=cnt,
            nsplit=-2, out_amax=amax[1:2], w_amax=wamax)


def plain():
    conv_ex(B=B, H=H, W=W, src=Act(x, cin, 0, cin, amax=amax[0:1]), w=wp, cout=cout, ks=ks, out=out, out_ctot=cout,
            out_coff=0, epi_mode=0, nsplit=-2, out_amax=amax[1:2], w_amax=wamax)


fl = 2.0 * B * H * W * cin * cout * ks * ks
for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("plain", plain)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"conv_ex {name:5s} {ks}x{ks} {cin}->{cout} B={B}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF algorithmic")
