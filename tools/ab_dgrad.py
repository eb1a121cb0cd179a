"""Same-box A/B of which bf16-storage dgrad launches run csrc/conv_b16k.hip (tsr_conv2d_ex nsplit = -3):
   python tools/ab_dgrad.py {all|masked|no64|off} [bench.py arguments]
'all' is the shipped choice; 'masked' keeps only the epi_mode-2 launches there; 'no64' keeps the 64-channel 3x3 / 5x5 layers
(forward and dgrad) on conv_mfma_split16; 'off' none (all on conv_mfma_split16)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
which = sys.argv.pop(1)
from tactilesr_amd import _lib
from tactilesr_amd.model import _train

lib = _lib.load()
real = lib.tsr_conv2d_ex_dgrad_b16k
state = {"masked": True}
orig = _train.TrainEngine._dgrad


def _dgrad(self, c, dz, conv, ci0, nprime, out, out_ctot, out_coff, res=None, mask=None, *a, **k):
    state["masked"] = mask is not None
    return orig(self, c, dz, conv, ci0, nprime, out, out_ctot, out_coff, res, mask, *a, **k)


def pred(n, c, ks):
    if which == "off" or (which == "masked" and not state["masked"]) or (which == "no64" and n == 64 and ks > 1):
        return 0
    return real(n, c, ks)


_train.TrainEngine._dgrad = _dgrad
lib.tsr_conv2d_ex_dgrad_b16k = pred
import bench
bench.main()
