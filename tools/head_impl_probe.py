import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import tactilesr_oracle as O
from tactilesr_amd.model import tactileSR_model as M
g = np.load("tests/golden/eval.npz")
cfg = dict(scale_factor=25, seqsCnt=8)
sd = O.random_state_dict(O.tactilesr_state_shapes(**cfg), int(g["sf25t8/seed"]))
LR = torch.from_numpy(g["sf25t8/LR"]).cuda()
def rel(a, b):
    a, b = a.detach().cpu().double(), b.double()
    return float((a - b).abs().max() / b.abs().max())
for head in (None, "bf16x6", "f32"):
    m = M.TactileSR(**cfg); m.load_state_dict(sd); m = m.cuda().eval(); m.conv_impl = "fp16x3"; m.head_impl = head
    y = m(LR)
    print("head_impl", head, "vs ref32 %.3e vs f64 %.3e" % (rel(y[0, 0, ::2, ::2], torch.from_numpy(g["sf25t8/out_full0"])), rel(y[0, 0, ::2, ::2], torch.from_numpy(g["sf25t8/out64_full0"]))))
