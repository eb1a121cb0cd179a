#!/usr/bin/env python3
"""Loss trajectory of the HIP train step against the CPU oracle's (tests/-style checker, not product code):
same seeded weights, same fixed batch, N Adam(L2) steps.  python tools/train_trajectory.py [B steps lr]
(at the reference default lr = 1e-3 without its warm-up scheduler the final ReLU of this net dies after one step on
synthetic targets -- on the CPU oracle exactly as here; use 1e-5..1e-4 to see a live trajectory)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tactilesr_oracle as O  # noqa: E402
import tactilesr_amd  # noqa: E402
from tactilesr_amd import optim  # noqa: E402
from tactilesr_amd.train import tactileSR_train as TR  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
cfg = dict(patternFeatureExtraLayerCnt=2)
sd = O.random_state_dict(O.tactilesr_state_shapes(**cfg), 42)
g = torch.Generator().manual_seed(1)
LR = torch.rand(B, 3, 4, 4, generator=g) * 8
HR = torch.nn.functional.interpolate(LR.mean(1, keepdim=True), size=(100, 100), mode="bilinear") * 30

p = {k: v.clone() for k, v in sd.items()}
state = {}
ref = []
torch.set_num_threads(min(16, os.cpu_count() or 1))
for it in range(steps):
    loss, _ = O.train_one_iter(p, state, it + 1, LR, HR, lr=lr, weight_decay=1e-2)
    ref.append(loss)

m = tactilesr_amd.TactileSR(**cfg)
m.load_state_dict(sd, strict=True)
m = m.cuda().train()
opt = optim.Adam(m.parameters(), lr=lr, weight_decay=1e-2)
conf = TR.default_config()
got = []
for it in range(steps):
    got.append(float(TR.train_one_iter(m, opt, (LR, HR), conf)["total_loss"]))
for i, (a, b) in enumerate(zip(ref, got)):
    print(f"step {i:2d}: oracle {a:12.6f}   hip {b:12.6f}   rel diff {abs(a - b) / max(abs(a), 1e-30):.2e}")
