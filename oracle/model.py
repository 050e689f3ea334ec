"""ORACLE (test infrastructure) — CPU restatement of the reference's simulate / diagnostics / entropy /
prior / loss / MENTFlow.loss chain.  Pinned against the reference's own code via
tests/golden/ref_*.npz (oracle/gen_golden.py).  Paths relative to /root/reference.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import kde as _kde
from . import flow as _flow


# ---------------------------------------------------------------- simulate
class LinearTransform:
    """mentflow/simulate/transform.py:58-75."""

    def __init__(self, matrix: torch.Tensor) -> None:
        self.matrix = matrix
        self.matrix_inv = torch.linalg.inv(matrix)                # :63-65

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return torch.matmul(x, self.matrix.T)                     # :67-68

    def inverse(self, u: torch.Tensor) -> torch.Tensor:
        return torch.matmul(u, self.matrix_inv.T)                 # :70-71


class MultipoleTransform:
    """mentflow/simulate/transform.py:78-146 (orders 3..5 are the ones that do not raise; note U[:,3] reads X[:,1])."""

    def __init__(self, order: int, strength: float, skew: bool = False) -> None:
        self.order, self.strength, self.skew = int(order), float(strength), bool(skew)

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        import math
        U = X.clone()
        x = X[:, 0]
        y = X[:, 2] if X.shape[1] > 2 else 0.0 * X[:, 0]
        if self.order == 3:
            re, im = x ** 2 - y ** 2, 2.0 * x * y
        elif self.order == 4:
            re, im = x ** 3 - 3.0 * y ** 2 * x, -y ** 3 + 3.0 * x ** 2 * y
        elif self.order == 5:
            re, im = x ** 4 - 6.0 * x ** 2 * y ** 2 + y ** 4, 4.0 * x ** 3 * y - 4.0 * x * y ** 3
        else:
            raise ValueError("MPS-compatible MultipoleTransform requires order <= 5.")
        k = self.strength / math.factorial(self.order - 1)
        if self.skew:
            U[:, 1] = X[:, 1] + k * im
            if X.shape[1] > 2:
                U[:, 3] = X[:, 3] + k * re
        else:
            U[:, 1] = X[:, 1] - k * re
            if X.shape[1] > 2:
                U[:, 3] = X[:, 1] + k * im
        return U


class CompositeTransform:
    """transform.py:35-55."""

    def __init__(self, *transforms) -> None:
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def make_transforms_2d_nonlinear(num: int = 4, max_angle: float = 90.0, order: int = 3, max_strength: float = 1.5):
    """experiments/rec_2d/nonlinear/setup.py:24-44."""
    import numpy as np
    out = []
    for strength in np.linspace(-max_strength, max_strength, num):
        rot = LinearTransform(rotation_matrix(np.radians(max_angle)).type(torch.float32))
        out.append(CompositeTransform(MultipoleTransform(order, float(strength)), rot))
    return out


def simulate_forward(x, transforms, diagnostics) -> List[List[torch.Tensor]]:
    """mentflow/simulate/simulate.py:8-33."""
    predictions = []
    for index, transform in enumerate(transforms):
        u = transform(x.clone())                                  # :32
        predictions.append([diagnostic(u) for diagnostic in diagnostics[index]])   # :33
    return predictions


# ---------------------------------------------------------------- diagnostics
class Histogram1D:
    """mentflow/diagnostics/diagnostics.py:71-131 (noise branch omitted: off during training,
    experiments/setup.py:58-60)."""

    def __init__(self, edges: torch.Tensor, bandwidth: Optional[float] = None, axis: int = 0,
                 direction: Optional[torch.Tensor] = None, kde: bool = True) -> None:
        self.axis, self.kde, self.ndim = axis, kde, 1
        self.direction = None if direction is None else direction / torch.norm(direction)   # :103-105
        if bandwidth is None:
            bandwidth = 0.5                                                                  # :107-108
        self.edges = edges
        self.coords = 0.5 * (edges[:-1] + edges[1:])                                          # utils/grid.py:5-6
        self.resolution = edges[1] - edges[0]                                                 # :112
        self.bandwidth = bandwidth * self.resolution                                          # :113

    def project(self, x):
        if self.direction is None:
            return x[:, self.axis]                                                           # :118-119
        return torch.sum(x * self.direction, dim=1)                                          # :121

    def __call__(self, x):
        xp = self.project(x)
        if self.kde:
            return _kde.kde_histogram_1d(xp, self.edges, bandwidth=self.bandwidth)           # :125-127
        return _kde.hard_histogram_1d(xp, self.edges)                                        # :128-131


class Histogram2D:
    """mentflow/diagnostics/diagnostics.py:134-201."""

    def __init__(self, axis: Sequence[int], edges: Sequence[torch.Tensor],
                 bandwidth: Sequence[Optional[float]] = (None, None), kde: bool = True) -> None:
        self.axis, self.kde, self.ndim = tuple(axis), kde, 2
        bx, by = bandwidth
        bx = 0.5 if bx is None else bx                                                       # :163-166
        by = 0.5 if by is None else by
        self.edges_x, self.edges_y = edges
        self.edges = (self.edges_x, self.edges_y)
        self.resolution_x = self.edges_x[1] - self.edges_x[0]
        self.resolution_y = self.edges_y[1] - self.edges_y[0]
        self.bandwidth_x = bx * self.resolution_x                                            # :174
        self.bandwidth_y = by * self.resolution_y                                            # :175

    def project(self, x):
        return x[:, self.axis]                                                               # :179-180

    def __call__(self, x):
        xp = self.project(x)
        if self.kde:
            return _kde.kde_histogram_2d(xp[:, 0], xp[:, 1], bins=(self.edges_x, self.edges_y),
                                         bandwidth=(self.bandwidth_x, self.bandwidth_y))    # :183-190
        return _kde.hard_histogram_2d(xp, self.edges_x, self.edges_y)                        # :191-201


def renormalize_measurement(measurement: torch.Tensor, diagnostic) -> torch.Tensor:
    """experiments/setup.py:63-73."""
    if measurement.ndim == 1:
        volume = diagnostic.edges[1] - diagnostic.edges[0]
    else:
        volume = math.prod([e[1] - e[0] for e in diagnostic.edges])
    return measurement / torch.sum(measurement) / volume


# ---------------------------------------------------------------- prior / entropy / loss
class GaussianPrior:
    """mentflow/prior.py:4-26: MultivariateNormal(0, scale^2 I).log_prob."""

    def __init__(self, ndim: int = 2, scale: float = 1.0, dtype=torch.float32) -> None:
        self.ndim, self.scale = ndim, scale
        loc = torch.zeros(ndim, dtype=dtype)
        cov = torch.eye(ndim, dtype=dtype) * (scale**2)
        self._dist = torch.distributions.MultivariateNormal(loc, cov)

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return self._dist.log_prob(x)


def mc_entropy(x: torch.Tensor, log_prob: torch.Tensor, prior=None) -> torch.Tensor:
    """mentflow/entropy.py:58-62 (MonteCarloEntropyEstimator.forward)."""
    H = torch.mean(log_prob)
    if prior is not None:
        H = H - torch.mean(prior.log_prob(x))
    return H


def mean_absolute_error(pred, targ):
    """mentflow/loss.py:7-8."""
    return torch.mean(torch.abs(pred - targ))


def mean_square_error(pred, targ):
    """mentflow/loss.py:11-12."""
    return torch.mean(torch.square(pred - targ))


def kl_divergence(pred: torch.Tensor, targ: torch.Tensor, pad: float = 1.0e-12) -> torch.Tensor:
    """mentflow/loss.py:15-17."""
    log_pred = torch.log(pred + pad)
    return torch.nn.functional.kl_div(log_pred, targ, reduction="batchmean")


def unravel(iterable):
    """mentflow/utils/utils.py:10-11."""
    return itertools.chain.from_iterable(iterable)


def mentflow_loss(x, log_prob, transforms, diagnostics, measurements, prior, penalty_parameter,
                  discrepancy_function: Callable = kl_divergence):
    """mentflow/core.py:84-117 with (x, log_prob) injected: returns (L, H, [D_p])."""
    if log_prob is None:                                                       # entropy.py:21-26 (EmptyEntropyEstimator)
        H = torch.zeros((), dtype=x.dtype)
    else:
        H = mc_entropy(x, log_prob, prior)                                     # :84-87
    predictions = simulate_forward(x, transforms, diagnostics)                 # :114
    D = [discrepancy_function(pred, meas)                                      # :89-93
         for pred, meas in zip(unravel(predictions), unravel(measurements))]
    L = H + penalty_parameter * (sum(D) / len(D))                              # :116
    return L, H, D


def train_step_loss(z, spec: "_flow.FlowSpec", transforms, diagnostics, measurements, prior,
                    penalty_parameter, discrepancy_function: Callable = kl_divergence):
    """One MENTFlow.loss() with the base draw z injected (core.py:95-117 + flows/zuko.py:24-26)."""
    x, logp = _flow.sample_and_log_prob(z, spec)
    L, H, D = mentflow_loss(x, logp, transforms, diagnostics, measurements, prior,
                            penalty_parameter, discrepancy_function)
    return L, H, D, x, logp


# ---------------------------------------------------------------- experiment plumbing (spec of the benchmark configs)
def make_directions(num: int, ndim: int, seed: Optional[int]) -> torch.Tensor:
    """experiments/rec_nd_1d/setup.py:28-37 ("isotropic" optics): CPU generator stream."""
    rng = torch.Generator(device="cpu")
    if seed is not None:
        rng.manual_seed(seed)
    directions = torch.randn((num, ndim), generator=rng, device="cpu")
    return directions / torch.norm(directions, dim=1)[:, None]


def make_transforms_nd_1d(num: int, ndim: int, seed: Optional[int]) -> List[LinearTransform]:
    """experiments/rec_nd_1d/setup.py:39-49: M = I with row 0 := direction."""
    transforms = []
    for direction in make_directions(num, ndim, seed):
        M = torch.eye(ndim)
        M[0, :] = direction
        transforms.append(LinearTransform(M.float()))
    return transforms


def rotation_matrix(angle: float) -> torch.Tensor:
    """mentflow/simulate/transform.py:12-15."""
    import numpy as np
    c, s = np.cos(angle), np.sin(angle)
    return torch.tensor([[c, s], [-s, c]])


def make_transforms_2d_linear(num: int, min_angle: float = 0.0, max_angle: float = 180.0):
    """experiments/rec_2d/linear/setup.py:27-43."""
    import numpy as np
    angles = np.linspace(np.radians(min_angle), np.radians(max_angle), num, endpoint=False)
    return [LinearTransform(rotation_matrix(a).type(torch.float32)) for a in angles]


def make_transforms_nd_2d_corner(ndim: int) -> List[LinearTransform]:
    """experiments/rec_nd_2d/setup.py:38-53 ("corner" optics): all axis pairs moved to axes (0, 2)."""
    transforms = []
    axis_meas = (0, 2)
    for i in range(ndim):
        for j in range(i):
            matrices = []
            for k, l in zip(axis_meas, (j, i)):
                matrix = torch.eye(ndim)
                matrix[k, k] = matrix[l, l] = 0.0
                matrix[k, l] = matrix[l, k] = 1.0
                matrices.append(matrix.float())
            transforms.append(LinearTransform(torch.linalg.multi_dot(matrices[::-1])))
    return transforms
