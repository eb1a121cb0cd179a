"""tactilesr_amd: MI355X-native (gfx950) implementation of the tactileSR hot path --
the TactileSR conv upscaler and the tPSFNet PSF forward model -- behind the reference's
own model interface.  Host code is Python on PyTorch-ROCm; every hot op is a hand-written
HIP kernel in ``lib/libtactilesr_hip.so`` reached through the C ABI of
``include/tactilesr_hip.h``.  No CPU fallback exists in this package.
"""
from . import _lib  # noqa: F401
from .model.tactileSR_model import TactileSR, TactileSRCNN, MSRB, ResBlock  # noqa: F401
from .model.tPSFNet import tPSFNet  # noqa: F401

__all__ = ["TactileSR", "TactileSRCNN", "MSRB", "ResBlock", "tPSFNet"]
