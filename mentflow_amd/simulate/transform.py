"""Transport maps — mirrors mentflow/simulate/transform.py (LinearTransform :58-75, rotation_matrix :12-15)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def rotation_matrix(angle: float) -> torch.Tensor:
    """transform.py:12-15."""
    _cos, _sin = np.cos(angle), np.sin(angle)
    return torch.tensor([[_cos, _sin], [-_sin, _cos]])


class Transform(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class LinearTransform(Transform):
    """u = x @ M.T (transform.py:58-75).

    Inside ``simulate.forward`` / ``MENTFlow.loss`` the matrix is never applied as a whole: the fused projection
    kernels read only the rows the diagnostic consumes (``matrix[axis]``).  ``forward``/``inverse`` exist for API
    parity (plots, notebooks); a plain [N,d]x[d,d] library GEMM is not part of the hot path."""

    def __init__(self, matrix: torch.Tensor) -> None:
        super().__init__()
        self.set_matrix(matrix)

    def set_matrix(self, matrix: torch.Tensor) -> None:
        self.matrix = matrix
        self.matrix_inv = torch.linalg.inv(matrix)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.matmul(x, self.matrix.T)

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        return torch.matmul(u, self.matrix_inv.T)

    def to(self, device):
        self.matrix = self.matrix.to(device)
        self.matrix_inv = self.matrix_inv.to(device)     # the reference forgets this one (transform.py:73-75)
        return self


def reverse_momentum(x):
    """transform.py:18-21, on a copy (the reference flips the momenta of its argument in place, so its
    MultipoleTransform.inverse(u) also changes the caller's u; the returned value is the same)."""
    x = x.clone()
    for i in range(0, x.shape[1], 2):
        x[:, i + 1] *= -1.0
    return x


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
    """transform.py:35-55."""

    def __init__(self, *transforms) -> None:
        super().__init__()
        self.transforms = nn.Sequential(*transforms)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        u = x
        for transform in self.transforms:
            u = transform(u)
        return u

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        x = u
        for transform in list(self.transforms)[::-1]:
            x = transform.inverse(x)
        return x

    def to(self, device):
        for transform in self.transforms:
            transform.to(device)
        return self
