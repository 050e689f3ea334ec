"""Generator interface (method names of mentflow/generate/base.py:8-26 and mentflow/types_.py:13-26).

What `MENTFlow`, `Trainer` and the experiment scripts call on a generator:

    sample(n) -> x[n, d]                       sample_and_log_prob(n) -> (x, log_prob or None)
    log_prob(x) -> [n] or None                 sample_base(n) -> z[n, d_base]
    forward(z) -> x      inverse(x) -> z       forward_steps(z) / inverse_steps(x) -> list of intermediate states
    dim() -> d
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


class GenerativeModel(torch.nn.Module):
    """Every method raises until a subclass provides it (flows: generate/flows.py; plain network: generate/nn.py)."""

    def _missing(self, what: str):
        return NotImplementedError(f"{type(self).__name__} does not implement {what}()")

    # ---- sampling
    def sample_base(self, size: int) -> torch.Tensor:
        raise self._missing("sample_base")

    def sample(self, size: int) -> torch.Tensor:
        raise self._missing("sample")

    def sample_and_log_prob(self, size: int) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        raise self._missing("sample_and_log_prob")

    # ---- density
    def log_prob(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        raise self._missing("log_prob")

    # ---- the map itself
    def forward(self, z: torch.Tensor, **kws) -> torch.Tensor:
        raise self._missing("forward")

    def inverse(self, x: torch.Tensor, **kws) -> torch.Tensor:
        raise self._missing("inverse")

    def forward_steps(self, z: torch.Tensor) -> List[torch.Tensor]:
        raise self._missing("forward_steps")

    def inverse_steps(self, x: torch.Tensor) -> List[torch.Tensor]:
        raise self._missing("inverse_steps")

    def dim(self) -> int:
        raise self._missing("dim")
