"""Transport maps — mirrors mentflow/simulate/transform.py (LinearTransform :58-75, rotation_matrix :12-15)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


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
