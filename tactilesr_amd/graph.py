"""HIP-graph replay of the eval forward for small batches (real-time tactile SR).

``GraphedTactileSR`` captures one eval forward of ``TactileSR`` -- ~37 launches, every one through the C ABI on
torch's current stream, which is the capturing stream inside ``torch.cuda.graph`` -- into a HIP graph with static
input/output buffers and replays it per call: one host call per frame instead of ~80 (launches + allocations), so
a real-time host loop is free between frames.  Measured (tools/latency_probe.py): it does NOT shorten the frame
latency -- 1.38 ms at B=1, 2.2 ms at B=32 either way -- because that latency is the serial depth of one
workgroup's K loop (25 workgroups at B=1), not the launch cost.

The reference has no counterpart (it calls ``model(LR)`` per batch, train/tactileSR_train.py:78-84); this is the
serving-side wrapper around the same module and weights, bit-identical to the eager path.
"""
from __future__ import annotations

import torch

from . import _lib


class GraphedTactileSR:
    """``g = GraphedTactileSR(model.eval(), batch); y = g(x)`` with ``x`` of shape ``(batch, 3*seqsCnt, 4, 4)``.

    ``y`` is a static buffer that the next call overwrites (clone it to keep it).  The captured graph holds the
    packed weights of the moment of capture: build a new wrapper after the weights change (``__call__`` raises if
    the module's weight plan was rebuilt)."""

    def __init__(self, model, batch: int, height: int = 4, width: int = 4):
        if model.training:
            raise _lib.TactileSRHipError("GraphedTactileSR captures the eval forward: call model.eval() first")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.TactileSRHipError("GraphedTactileSR needs the model on an MI355X device")
        self.model, self.batch = model, int(batch)
        self.x = torch.zeros(self.batch, model.seqsCnt * model.axisCnt, height, width, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                 # builds the packed-weight plan and warms the allocator outside the capture
                model(self.x)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._plan = model._plan
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.y = model(self.x)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.model._plan is not self._plan:
            raise _lib.TactileSRHipError("the model's weights changed since capture: build a new GraphedTactileSR")
        if tuple(x.shape) != tuple(self.x.shape):
            raise AssertionError(f"captured for input {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.y
