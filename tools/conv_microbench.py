#!/usr/bin/env python3
"""Time one eval-mode convolution launch alone (random data, the network's shapes):

    python tools/conv_microbench.py [ks cin cout B impl]      impl: fp16x3 (default) | bf16 (bf16 storage, one plane)

Prints ms per launch and the algorithmic / executed TFLOP/s.  A/B a kernel experiment with
TSR_LIB_OVERRIDE=tactilesr_amd/lib/exp/NAME/libtactilesr_hip.so (tools/build_variant.py)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, c_int as I, c_float as F  # noqa: E402

argv = sys.argv[1:]
ks, cin, cout, B = [int(x) for x in (argv[:4] + ["5", "128", "128", "4096"][len(argv[:4]):])]
impl = argv[4] if len(argv) > 4 else "fp16x3"
H = W = 40
g = torch.Generator().manual_seed(0)
x = torch.randn(B * cin * H * W, generator=g).clamp_(min=0).cuda()          # post-ReLU-like activations
w = (torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
scale = torch.ones(cout, device="cuda")
shift = torch.zeros(cout, device="cuda")
amax = torch.stack([x.abs().max(), torch.zeros((), device="cuda")]).contiguous()
if impl == "fp16x3":
    out = torch.empty(B * cout * H * W, device="cuda")
    wp = torch.empty(2 * cout * cin * (ks * ks + 1), dtype=torch.float16, device="cuda")
    wscale = 2.0 ** (13 - int(torch.floor(torch.log2(w.abs().max()))))
    call("tsr_pack_conv_weight_f16s", ptr(w), ptr(wp), I(cout), I(cin), I(ks), F(wscale), stream())

    def run():
        call("tsr_conv2d_fwd_f16s", ptr(x), I(cin), I(0), I(cin), ptr(wp), I(cout), I(ks), F(1.0 / wscale),
             ptr(amax[0:1]), ptr(amax[1:2]), ptr(scale), ptr(shift), ptr(None), I(0), I(0), ptr(out), I(cout), I(0), I(1),
             I(B), I(H), I(W), stream())
    nprod = 3
else:
    xb = x.to(torch.bfloat16)
    out = torch.empty(B * cout * H * W, dtype=torch.bfloat16, device="cuda")
    wp = torch.empty(cout * cin * (ks * ks + 1), dtype=torch.bfloat16, device="cuda")
    call("tsr_pack_conv_weight_bf16s", ptr(w), ptr(wp), I(cout), I(cin), I(ks), I(1), stream())

    def run():
        call("tsr_conv2d_fwd_b16", ptr(xb), I(cin), I(0), I(cin), ptr(wp), I(cout), I(ks), ptr(scale), ptr(shift),
             ptr(None), I(0), I(0), ptr(out), I(cout), I(0), I(1), I(B), I(H), I(W), stream())
    nprod = 1

for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * H * W * cin * cout * ks * ks
print(f"conv {ks}x{ks} {cin}->{cout} B={B} {impl}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF algorithmic, "
      f"{fl * nprod / ms / 1e9:.0f} TF executed ({fl * nprod / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)  "
      f"checksum {float(out.float().abs().mean()):.6g}")
