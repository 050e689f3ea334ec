"""ORACLE (test infrastructure) — CPU restatement of the reference's differentiable KDE histograms
and hard-binned histograms.  Pinned against the reference's own code via tests/golden/ref_*.npz.

Follows /root/reference/mentflow/diagnostics/histogram.py (line numbers cited per function) and
mentflow/diagnostics/diagnostics.py:124-131,182-201.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import numpy as np
import torch


def marginal_pdf(values: torch.Tensor, coords: torch.Tensor, sigma, epsilon: float = 1.0e-10):
    """histogram.py:11-44.  values [n,1], coords [k] -> (prob [k], kernel_values [n,k]).
    Dense: materialises the n x k kernel matrix (exactly what the reference does)."""
    residuals = values - coords.repeat(*values.shape)                 # :37
    kernel_values = torch.exp(-0.5 * (residuals / sigma).pow(2))      # :38
    prob = torch.mean(kernel_values, dim=-2)                          # :39
    delta = coords[1] - coords[0]                                     # :40
    normalization = torch.sum(prob * delta) + epsilon                 # :41-42
    return prob / normalization, kernel_values                        # :43-44


def joint_pdf(kx: torch.Tensor, ky: torch.Tensor, coords, epsilon: float = 1.0e-10) -> torch.Tensor:
    """histogram.py:47-74.  prob = Kx^T Ky (NOT divided by n), normalised by sum*dx*dy + eps."""
    prob = torch.matmul(kx.transpose(-2, -1), ky)                     # :69
    delta = [c[1] - c[0] for c in coords]                             # :70
    normalization = torch.sum(prob * delta[0] * delta[1]) + epsilon   # :71-72
    return prob / normalization                                       # :73


def kde_histogram_1d(x: torch.Tensor, bins: torch.Tensor, bandwidth=1.0, epsilon: float = 1.0e-10):
    """histogram.py:77-86."""
    coords = 0.5 * (bins[:-1] + bins[1:])
    prob, _ = marginal_pdf(x.unsqueeze(-1), coords, bandwidth, epsilon)
    return prob


def kde_histogram_2d(x, y, bins: Iterable[torch.Tensor], bandwidth=(1.0, 1.0), epsilon: float = 1.0e-10):
    """histogram.py:89-101."""
    coords = [0.5 * (e[:-1] + e[1:]) for e in bins]
    _, kx = marginal_pdf(x.unsqueeze(-1), coords[0], bandwidth[0], epsilon)
    _, ky = marginal_pdf(y.unsqueeze(-1), coords[1], bandwidth[1], epsilon)
    return joint_pdf(kx, ky, coords, epsilon=epsilon)


def hard_histogram_1d(x_proj: torch.Tensor, edges: torch.Tensor) -> torch.Tensor:
    """diagnostics.py:128-131: torch.histogram(x_proj, edges, density=True).hist."""
    return torch.histogram(x_proj, edges, density=True).hist


def hard_histogram_2d(x_proj: torch.Tensor, edges_x: torch.Tensor, edges_y: torch.Tensor) -> torch.Tensor:
    """diagnostics.py:191-201: np.histogramdd(density=True) -> float32 tensor."""
    hist, _ = np.histogramdd(
        x_proj.detach().cpu().numpy(),
        bins=[edges_x.detach().cpu().numpy(), edges_y.detach().cpu().numpy()],
        density=True,
    )
    return torch.from_numpy(hist).type(torch.float32)
