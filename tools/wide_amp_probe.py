#!/usr/bin/env python3
"""Per-frame error table of the wide-amplitude B=4096 batch (tests/test_gpu_parity.py) -> gpurun_out/wide_amp.npz"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import tactilesr_oracle as O
import tactilesr_amd
g = np.load(os.path.join(REPO, "tests/golden/eval_init.npz"))
torch.manual_seed(42)
m = tactilesr_amd.TactileSR()
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
for k in sd:
    if f"init_t1/stat/{k}" in g.files:
        sd[k] = torch.from_numpy(g[f"init_t1/stat/{k}"])
m.load_state_dict(sd)
m = m.cuda().eval()
gen = torch.Generator().manual_seed(2024)
base = torch.rand(256, 3, 4, 4, generator=gen) * 8
base = base * (2.0 ** -(torch.arange(256) % 9).float()).view(-1, 1, 1, 1)
with torch.no_grad():
    ref = O.tactilesr_forward(sd, base).double()
    ref64 = O.tactilesr_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, base.double())
out = {"fmax": ref64.abs().amax(dim=(1, 2, 3)).numpy(), "yard_abs": (ref - ref64).abs().amax(dim=(1, 2, 3)).numpy()}
for impl in ("fp16x3", "f32", "bf16x6"):
    m.conv_impl = impl
    y = m(base.repeat(16, 1, 1, 1).cuda())[:256].cpu().double()
    out[f"{impl}/abs32"] = (y - ref).abs().amax(dim=(1, 2, 3)).numpy()
    out[f"{impl}/abs64"] = (y - ref64).abs().amax(dim=(1, 2, 3)).numpy()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(REPO, "gpurun_out/wide_amp.npz"), **out)
print("ok")
