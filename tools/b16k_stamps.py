#!/usr/bin/env python3
"""Single-launch timings of conv_b16k_kernel (fused 3x3 / 5x5, pair, plain 3x3 128) at B = 4096, random data.  Against a
library that exports tsr_debug_set_stamps -- the diagnostic -DTSR_STAMP build, which lives in the history of
csrc/conv_b16k.hip at commit 9707380 (per-workgroup s_memtime stamps at kernel start, end of prologue, end of main loop,
end of epilogue; it is not in the shipping source) -- it also prints the per-workgroup median of the three phases:

    git worktree add /tmp/wt 9707380 && (cd /tmp/wt && python tools/build_variant.py stamp -DTSR_STAMP)
    TSR_ALLOW_VARIANT=1 TSR_LIB_OVERRIDE=/tmp/wt/tactilesr_amd/lib/exp/stamp/libtactilesr_hip.so python tools/b16k_stamps.py
"""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactilesr_amd._lib import call, ptr, stream, load, c_int as I  # noqa: E402

B, H, W = 4096, 40, 40
lib = load()
HAVE_STAMPS = hasattr(lib, "tsr_debug_set_stamps")          # the -DTSR_STAMP variant only; otherwise: timing only
if HAVE_STAMPS:
    lib.tsr_debug_set_stamps.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(0)


def run(kind, ks, cin):
    x = torch.randn(B * cin * H * W, generator=g).clamp_(min=0).to(torch.bfloat16).cuda()
    w = (torch.randn(128, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5).cuda()
    wp = torch.empty(lib.tsr_conv_weight_b16k_elems(128, cin, ks), dtype=torch.bfloat16, device="cuda")
    call("tsr_pack_conv_weight_b16k", ptr(w), ptr(wp), I(128), I(cin), I(ks), stream())
    sc, sh = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    nwg = (B // 4) * 25
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    if kind == "fused":
        w2 = (torch.randn(64, 128, generator=g) * 0.1).cuda()
        w2p = torch.empty(64 * 128, dtype=torch.bfloat16, device="cuda")
        call("tsr_pack_w2_b16k", ptr(w2), ptr(w2p), stream())
        res = torch.randn(B * 64 * H * W, generator=g).to(torch.bfloat16).cuda()
        out = torch.empty(B * 64 * H * W, dtype=torch.bfloat16, device="cuda")

        def go():
            call("tsr_conv2d_fwd_b16k_fuse1x1", ptr(x), I(cin), I(0), I(cin), ptr(wp), I(ks), ptr(sc), ptr(sh), I(1), ptr(w2p),
                 ptr(None), ptr(res), I(64), I(0), ptr(out), I(64), I(0), I(1), I(B), I(H), I(W), stream())
    elif kind == "pair":
        out = torch.empty(B * 128 * H * W, dtype=torch.bfloat16, device="cuda")

        def go():
            call("tsr_conv2d_fwd_b16k_pair", ptr(x), I(cin), I(0), I(cin), ptr(wp), ptr(sc), ptr(sh), ptr(out), I(128), I(0),
                 I(1), I(B), I(H), I(W), stream())
    else:
        out = torch.empty(B * 128 * H * W, dtype=torch.bfloat16, device="cuda")

        def go():
            call("tsr_conv2d_fwd_b16k", ptr(x), I(cin), I(0), I(cin), ptr(wp), I(128), I(ks), ptr(sc), ptr(sh), ptr(None), I(0),
                 I(0), ptr(out), I(128), I(0), I(1), I(B), I(H), I(W), stream())
    if HAVE_STAMPS:
        lib.tsr_debug_set_stamps(None)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if not HAVE_STAMPS:
        print(f"{kind} k{ks} cin{cin}: {ms:.3f} ms/launch")
        return
    lib.tsr_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    go()
    torch.cuda.synchronize()
    lib.tsr_debug_set_stamps(None)
    t = stamps.view(nwg, 8)[:, :4].cpu().double()
    d = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
    med = d.median(0).values
    # s_memtime ticks at 100 MHz on gfx950 (s_memrealtime) or at the shader clock (s_memtime): report raw ticks
    span = float(t[:, 3].max() - t[:, 0].min())
    print(f"{kind} k{ks} cin{cin}: {ms:.3f} ms/launch; per-WG median ticks: prologue {med[0]:.0f}, main loop {med[1]:.0f}, "
          f"epilogue {med[2]:.0f}, total {med[3]:.0f}; launch span {span:.0f} ticks")


run("fused", 5, 128)
run("fused", 3, 128)
run("pair", 5, 64)
run("plain", 3, 128)
