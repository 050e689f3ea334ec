"""Generator interface — mirrors mentflow/generate/base.py:8-26 and mentflow/types_.py:13-26."""
from __future__ import annotations

from typing import List, Tuple

import torch


class GenerativeModel(torch.nn.Module):
    """Base class for generative models (same method names as the reference)."""

    def sample(self, size: int) -> torch.Tensor:
        raise NotImplementedError

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def sample_and_log_prob(self, size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def forward(self, z: torch.Tensor, **kws) -> torch.Tensor:
        raise NotImplementedError

    def inverse(self, x: torch.Tensor, **kws) -> torch.Tensor:
        raise NotImplementedError

    def forward_steps(self, z: torch.Tensor) -> List[torch.Tensor]:
        raise NotImplementedError

    def inverse_steps(self, x: torch.Tensor) -> List[torch.Tensor]:
        raise NotImplementedError

    def sample_base(self, size: int) -> torch.Tensor:
        raise NotImplementedError

    def dim(self) -> int:
        raise NotImplementedError
