"""hipGraph capture of the whole training step.

At the reference's own batch size (25 000 particles, experiments/rec_nd_1d/run_rings.sh:21) one step is ~45 kernel
launches of a few microseconds each: launch- and host-bound (1.2 ms/step eager on MI355X, of which 0.65 ms is GPU work).
`GraphedTrainStep` captures zero_grad + MENTFlow.loss + backward + optimizer.step once (torch.cuda.CUDAGraph = hipGraph on
ROCm; every mentflow_amd kernel is launched on torch's current stream, so it is captured like any torch op) and replays
it per iteration.  The base draw z ~ N(0, I) inside the graph uses torch's graph-safe Philox generator, so every replay
samples fresh particles.  Every kernel on the path is deterministic, so a replay reproduces the eager step bit for bit
on an injected base draw (tests/test_graph_capture.py).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def _copy_all(dst: List[torch.Tensor], src: List[torch.Tensor]) -> None:
    """dst[i] <- src[i]: one multi-tensor kernel where torch has it, a loop otherwise."""
    foreach = getattr(torch, "_foreach_copy_", None)
    if foreach is not None:
        foreach(dst, src)
    else:
        for d, s in zip(dst, src):
            d.copy_(s)


class GraphedTrainStep:
    """step() -> (L, H, mean D) as device tensors (static buffers, overwritten by the next replay).

    The optimizer must be capturable (e.g. ``torch.optim.AdamW(..., capturable=True)``); ``model.penalty_parameter`` and
    the learning rate are read at capture time — call ``recapture()`` after changing them (once per epoch in the
    penalty method).  The eager warm-up steps that graph capture needs are UNDONE afterwards (parameters and optimizer
    state are restored in place), so capturing does not consume training iterations.

    ``guard=True`` records a copy of the parameters and of the optimizer state at the head of every replay;
    ``undo_last_step()`` puts them back — the graph-mode equivalent of the reference skipping backward + step when the
    loss is not finite (mentflow/train/train.py:167-169)."""

    def __init__(self, model, optimizer, batch_size: int, warmup: int = 3, guard: bool = False) -> None:
        if warmup < 1:
            raise ValueError("at least one eager warm-up step is needed: the optimizer state must exist before capture")
        self.model, self.optimizer, self.batch_size = model, optimizer, int(batch_size)
        self.warmup = warmup
        self.guard = guard
        self.graph = None
        self._backup: Optional[List[torch.Tensor]] = None
        self.recapture()

    # ------------------------------------------------------------------ state handling (all in place: graph-stable)
    def _state_tensors(self) -> List[torch.Tensor]:
        out = [p for p in self.model.parameters()]
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                for v in self.optimizer.state.get(p, {}).values():
                    if torch.is_tensor(v):
                        out.append(v)
        return out

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=False)
        L, H, D = self.model.loss(self.batch_size)
        L.backward()
        self.optimizer.step()
        Hd = H.detach() if torch.is_tensor(H) else torch.zeros_like(L.detach())
        Dm = torch.stack([d.detach() for d in D]).mean()
        return L.detach(), Hd, Dm, torch.stack([L.detach(), Hd, Dm])

    def recapture(self) -> None:
        # (gradient tensors are created by the warm-up steps below: the flow generator installs views of ONE flat gradient
        # buffer, which is what keeps the captured accumulation a single kernel)
        before = self._state_tensors()
        saved = [t.detach().clone() for t in before]
        # warm-up on a side stream (allocator, lazy optimizer state), as torch.cuda.graphs requires
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        # undo the warm-up: tensors that existed before get their values back, optimizer state created by the warm-up
        # goes back to its initial value (zeros)
        known = {id(t) for t in before}
        with torch.no_grad():
            for t, v in zip(before, saved):
                t.copy_(v)
            for t in self._state_tensors():
                if id(t) not in known:
                    t.zero_()
        state = self._state_tensors()
        if self.guard:
            self._backup = [torch.empty_like(t) for t in state]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if self.guard:
                _copy_all(self._backup, state)
            self.out = self._eager()
        self._state = state

    def step(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        self.graph.replay()
        return self.out[:3]

    def scalars(self) -> torch.Tensor:
        """[L, H, mean D] of the last replay as ONE device tensor (a single device->host copy fetches all three)."""
        return self.out[3]

    def undo_last_step(self) -> None:
        if not self.guard:
            raise RuntimeError("GraphedTrainStep(guard=True) is needed to undo a step")
        with torch.no_grad():
            _copy_all(self._state, self._backup)
