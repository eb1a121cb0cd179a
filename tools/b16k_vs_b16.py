#!/usr/bin/env python3
"""Time one bf16-storage inference conv launch on both kernels (conv_b16k.hip / the 32x32x16 kernel):
    python tools/b16k_vs_b16.py ks cin cout B H W"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, load, c_int as I
ks, cin, cout, B, H, W = [int(x) for x in sys.argv[1:7]]
lib = load()
g = torch.Generator().manual_seed(0)
x = torch.randn(B * cin * H * W, generator=g).cuda().to(torch.bfloat16)
w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).cuda()
sc, sh = torch.rand(cout).cuda() + 0.5, torch.randn(cout).cuda() * 0.1
out = torch.empty(B * cout * H * W, dtype=torch.bfloat16, device="cuda")
wk = torch.empty(lib.tsr_conv_weight_b16k_elems(cout, cin, ks), dtype=torch.bfloat16, device="cuda")
call("tsr_pack_conv_weight_b16k", ptr(w), ptr(wk), I(cout), I(cin), I(ks), stream())
wo = torch.empty(lib.tsr_conv_weight_bf16s_elems(cout, cin, ks, 1), dtype=torch.bfloat16, device="cuda")
call("tsr_pack_conv_weight_bf16s", ptr(w), ptr(wo), I(cout), I(cin), I(ks), I(1), stream())
def run(name, wp):
    def f():
        call(name, ptr(x), I(cin), I(0), I(cin), ptr(wp), I(cout), I(ks), ptr(sc), ptr(sh), None, I(0), I(0), ptr(out), I(cout),
             I(0), I(1), I(B), I(H), I(W), stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * H * W * cin * cout * ks * ks
    print(f"{name} k{ks} {cin}->{cout} B={B} {H}x{W}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF  checksum {float(out.float().abs().mean()):.5f}")
run("tsr_conv2d_fwd_b16k", wk)
run("tsr_conv2d_fwd_b16", wo)
