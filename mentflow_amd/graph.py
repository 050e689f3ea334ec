"""hipGraph capture of the whole training step.

At the reference's own batch size (25 000 particles, experiments/rec_nd_1d/run_rings.sh:21) one step is ~50 kernel
launches of a few microseconds each: launch- and host-bound (2.8 ms/step eager on MI355X, of which < 0.5 ms is GPU work).
`GraphedTrainStep` captures zero_grad + MENTFlow.loss + backward + optimizer.step once (torch.cuda.CUDAGraph = hipGraph on
ROCm; every mentflow_amd kernel is launched on torch's current stream, so it is captured like any torch op) and replays
it per iteration.  The base draw z ~ N(0, I) inside the graph uses torch's graph-safe Philox generator, so every replay
samples fresh particles.
"""
from __future__ import annotations

from typing import Tuple

import torch


class GraphedTrainStep:
    """step() -> (L, H, mean D) as device tensors (static buffers, overwritten by the next replay).

    The optimizer must be capturable (e.g. ``torch.optim.AdamW(..., capturable=True)``); ``model.penalty_parameter``
    is read at capture time — call ``recapture()`` after changing it (once per epoch in the penalty method)."""

    def __init__(self, model, optimizer, batch_size: int, warmup: int = 3) -> None:
        if warmup < 1:
            raise ValueError("at least one eager warm-up step is needed: the optimizer state must exist before capture")
        self.model, self.optimizer, self.batch_size = model, optimizer, int(batch_size)
        self.warmup = warmup
        self.graph = None
        self.recapture()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=False)
        L, H, D = self.model.loss(self.batch_size)
        L.backward()
        self.optimizer.step()
        return L.detach(), (H.detach() if torch.is_tensor(H) else torch.zeros_like(L)), torch.stack([d.detach() for d in D]).mean()

    def recapture(self) -> None:
        # warm-up on a side stream (allocator, lazy module state), as torch.cuda.graphs requires
        for p in self.model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._eager()

    def step(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        self.graph.replay()
        return self.out
