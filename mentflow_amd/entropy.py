"""Entropy estimators — mirrors mentflow/entropy.py:8-62."""
from typing import Any, Optional

import torch

from . import ops
from .prior import Gaussian


class EntropyEstimator(torch.nn.Module):
    """Negative-entropy estimate H from samples and / or their log-density (interface of entropy.py:8-15); `prior`
    turns it into the relative entropy against that prior."""

    def __init__(self, prior: Any = None) -> None:
        super().__init__()
        self.prior = prior

    def forward(self, x: torch.Tensor, log_prob: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError(type(self).__name__)


class EmptyEntropyEstimator(EntropyEstimator):
    """No entropy term (entropy.py:18-24): the python float 0.0, whatever the inputs."""

    def forward(self, x: torch.Tensor, log_prob: Optional[torch.Tensor] = None) -> float:
        return 0.0


class MonteCarloEntropyEstimator(EntropyEstimator):
    """H = mean(log_prob) - mean(prior.log_prob(x))  (entropy.py:53-62), both means from one reduction kernel.

    ``from_sums`` finishes the estimate from already reduced sums — used by ``MENTFlow.loss`` so that a data-parallel
    run can sum [sum log_prob, sum |x|^2] across GPUs first."""

    def from_sums(self, sums: torch.Tensor, n_total: int) -> torch.Tensor:
        H = sums[0] / n_total
        if self.prior is not None:
            if not isinstance(self.prior, Gaussian):
                raise NotImplementedError("only the Gaussian prior is on the hot path (SURVEY.md §2 row 7)")
            H = H - (-0.5 * sums[1] / (n_total * self.prior.scale ** 2) + self.prior.log_norm())
        return H

    def forward(self, x: torch.Tensor, log_prob: torch.Tensor) -> torch.Tensor:
        return self.from_sums(ops.EntropySumsFn.apply(x, log_prob), x.shape[0])
