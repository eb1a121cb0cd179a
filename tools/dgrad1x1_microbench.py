#!/usr/bin/env python3
"""Time the streaming masked dgrad of the 1x1 confusion (csrc/conv1x1_b16k.hip) alone at B = 2048:
    python tools/dgrad1x1_microbench.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tactilesr_amd.model._train import conv_ex, Act, _pack_dgrad
from tactilesr_amd._lib import load
from tactilesr_amd.model import tactileSR_model as T
B, H, W = 2048, 40, 40
lib = load()
g = torch.Generator().manual_seed(0)
dy = torch.randn(B * 64 * H * W, generator=g).cuda().to(torch.bfloat16)
z = torch.randn(B * 256 * H * W, generator=g).cuda().to(torch.bfloat16)
w = (torch.randn(64, 256, 1, 1, generator=g) * 0.05).cuda()
v = lambda: (torch.rand(128) + 0.5).cuda()
mk = Act(z, 256, 0, 128, v(), v() - 1, v(), v() - 1)
out = torch.empty(B * 256 * H * W, dtype=torch.bfloat16, device="cuda")
entries = lib.tsr_conv2d_slab_entries_ex(B, H, W, 128, 1, -3)
slab = torch.empty(entries * 128 * 2, device="cuda")
wp = _pack_dgrad(w, 64, 256, 1, 0, 128, -3)
def f():
    conv_ex(B=B, H=H, W=W, src=Act(dy, 64, 0, 64), w=wp, cout=128, ks=1, out=out, out_ctot=256, out_coff=0, epi_mode=2, mask=mk, bn=True, slab=slab, nsplit=-3)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"1x1 dgrad B={B}: {ms:.3f} ms  {2.1e9 * B / 2048 / ms / 1e9:.2f} TB/s")
