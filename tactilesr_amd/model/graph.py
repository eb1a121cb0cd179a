"""HIP-graph replay of the eval forward for small batches.

The reference evaluates with ``test_batch_size = 8`` (config/default.py:53, train/tactileSR_train.py:66-101): at that
size the forward is ~45 launches of a few microseconds each and the HOST's launch path, not the GPU, sets the
latency.  ``GraphedForward`` captures one eval forward of a ``TactileSR`` (fixed input shape) into a HIP graph
(``torch.cuda.CUDAGraph``: the ctypes launches of the C ABI go to the capturing stream like any other kernel) and
replays it with one host call.  Opt-in, eval mode only; the plain ``model(x)`` path is untouched.

    g = GraphedForward(model, example)      # model.eval(); example: (B, 3T, 4, 4) on the ROCm device
    y = g(x)                                # same shape as `example`; y is the graph's OUTPUT BUFFER: clone it to keep it
"""
from __future__ import annotations

import torch

from .. import _lib


class GraphedForward:
    def __init__(self, model, example: torch.Tensor, warmup: int = 2):
        if model.training:
            raise _lib.TactileSRHipError("GraphedForward replays the EVAL forward: call model.eval() first")
        if not example.is_cuda:
            raise _lib.TactileSRHipError("GraphedForward needs a ROCm tensor (no CPU fallback)")
        self.model = model
        self.x = example.detach().float().contiguous().clone()
        self.warmup = int(warmup)
        self.captures = 0
        self._capture()

    def _capture(self) -> None:
        m = self.model
        # warm-up on a side stream (weight packs are built here, outside the capture), then record one forward
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, self.warmup)):
                m(self.x)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.y = m(self.x)
        self._plan = m._plan                 # the graph's kernels read these packed weights: keep them alive
        self._key = m._param_key()
        self.captures += 1

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        m = self.model
        if m.training:
            raise _lib.TactileSRHipError("GraphedForward replays the EVAL forward: the model is in train mode")
        if tuple(x.shape) != tuple(self.x.shape):
            raise _lib.TactileSRHipError(f"GraphedForward was captured for input shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        if m._param_key() != self._key:      # parameters / running statistics / conv_impl changed: record again
            self._capture()
        self.x.copy_(x)
        self.graph.replay()
        return self.y
