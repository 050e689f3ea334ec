"""Transport maps — mirrors mentflow/simulate/transform.py (LinearTransform :58-75, rotation_matrix :12-15)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def rotation_matrix(angle: float) -> torch.Tensor:
    """2 x 2 phase-space rotation [[c, s], [-s, c]] in float64, like transform.py:12-15 (numpy scalars)."""
    c, s = float(np.cos(angle)), float(np.sin(angle))
    return torch.tensor([[c, s], [-s, c]], dtype=torch.float64)


class Transform(nn.Module):
    """Interface of a transport map: ``forward`` (x -> u) and ``inverse`` (u -> x)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(f"{type(self).__name__}.forward")

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(f"{type(self).__name__}.inverse")


class LinearTransform(Transform):
    """u = x @ M.T (transform.py:58-75).

    Inside ``simulate.forward`` / ``MENTFlow.loss`` the matrix is never applied as a whole: the fused projection
    kernels read only the rows the diagnostic consumes (``matrix[axis]``).  ``forward``/``inverse`` exist for API
    parity (plots, notebooks); a plain [N,d]x[d,d] library GEMM is not part of the hot path."""

    def __init__(self, matrix: torch.Tensor) -> None:
        super().__init__()
        self.matrix = self.matrix_inv = None
        self.set_matrix(matrix)

    def set_matrix(self, matrix: torch.Tensor) -> None:
        self.matrix, self.matrix_inv = matrix, torch.linalg.inv(matrix)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x @ self.matrix.T

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        return u @ self.matrix_inv.T

    def to(self, device):
        # plain attributes, not buffers (the reference moves `matrix` only and leaves `matrix_inv` behind)
        self.matrix, self.matrix_inv = self.matrix.to(device), self.matrix_inv.to(device)
        return self


def reverse_momentum(x: torch.Tensor) -> torch.Tensor:
    """Momenta (odd columns) negated, on a copy.  transform.py:18-21 flips them in place, so the reference's
    MultipoleTransform.inverse(u) also changes the caller's u; the returned value is the same."""
    out = x.clone()
    out[:, 1::2].neg_()
    return out


class MultipoleTransform(Transform):
    """Thin multipole kick (transform.py:78-146); orders 3..5 (the reference's orders 1-2 fall into its `else: raise`).
    order: 3 = sextupole-like term z^2, ...;  strength: integrated kick;  skew: 45-degree rotated magnet."""

    def __init__(self, order: int, strength: float, skew: bool = False) -> None:
        super().__init__()
        if not 3 <= int(order) <= 5:
            raise ValueError("MPS-compatible MultipoleTransform requires order <= 5.")      # transform.py:131-132
        self.order = int(order)
        self.strength = float(strength)
        self.skew = bool(skew)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        k = self.strength / math.factorial(self.order - 1)                                    # transform.py:134
        return ops.MultipoleKickFn.apply(X, self.order, k, self.skew)

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        return reverse_momentum(self.forward(reverse_momentum(u)))                             # transform.py:145-146


class CompositeTransform(Transform):
    """Chain of transforms applied left to right (transform.py:35-55); ``transforms`` is an ``nn.Sequential`` like the
    reference's, so ``len`` / indexing / iteration work the same.  ``simulate.forward`` peels a trailing LinearTransform
    off the chain and fuses it into the projection kernel."""

    def __init__(self, *transforms) -> None:
        super().__init__()
        self.transforms = nn.Sequential(*transforms)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for stage in self.transforms:
            x = stage(x)
        return x

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        for stage in reversed(list(self.transforms)):
            u = stage.inverse(u)
        return u

    def to(self, device):
        for stage in self.transforms:
            stage.to(device)
        return self
