"""Prior distributions (interface of mentflow/prior.py:4-26)."""
import math

import torch


class Gaussian:
    """Isotropic Gaussian prior N(0, scale^2 I).  ``log_prob`` is the closed form of
    ``MultivariateNormal(0, scale^2 I).log_prob`` (prior.py:25-26); inside ``MonteCarloEntropyEstimator`` only its
    sufficient statistic sum|x|^2 is needed, which the entropy kernel reduces."""

    def __init__(self, ndim: int = 2, scale: float = 1.0, device=None) -> None:
        self.ndim, self.scale, self.device = int(ndim), float(scale), device

    def to(self, device) -> "Gaussian":
        self.device = device
        return self

    def log_norm(self) -> float:
        """log of the normalisation constant (2 pi scale^2)^(-ndim/2)."""
        return -self.ndim * (math.log(self.scale) + 0.5 * math.log(2.0 * math.pi))

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return self.log_norm() - 0.5 * x.square().sum(dim=1) / self.scale ** 2
