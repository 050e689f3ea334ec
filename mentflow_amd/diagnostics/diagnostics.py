"""Histogram diagnostics — mirrors mentflow/diagnostics/diagnostics.py:18-201 on the gfx950 kernels.

``Histogram1D`` / ``Histogram2D`` keep the reference's constructor kwargs, buffers (``edges``, ``coords``,
``resolution``, ``bandwidth``; ``edges_x`` ...) and the externally toggled attributes ``kde`` / ``noise``
(experiments/setup.py:52-60).  ``forward(u)`` evaluates ONE projection; ``simulate.forward`` batches all the
projections of a measurement set into a single fused kernel launch (``Histogram*.batched``).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple, Union

import torch

from .. import ops
from ..utils import coords_from_edges


class Diagnostic(torch.nn.Module):
    def __init__(self, device: torch.device = None, seed: int = None, ndim: int = None) -> None:
        super().__init__()
        self.device = device
        self.seed = seed
        self.ndim = ndim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class Histogram(Diagnostic):
    def __init__(self, noise: bool = False, noise_scale: float = 0.0, noise_type: str = "gaussian", **kws) -> None:
        super().__init__(**kws)
        self.noise = noise
        self.noise_scale = noise_scale
        self.noise_type = noise_type

    def set_noise(self, setting: bool) -> None:
        self.noise = setting

    def _apply_noise(self, hist: torch.Tensor) -> torch.Tensor:
        """diagnostics.py:53-67 (multiplicative measurement noise; off during training)."""
        if self.noise and self.noise_scale > 0.0:
            rng = torch.Generator(device=hist.device)
            if self.seed is not None:
                rng.manual_seed(self.seed)
            if self.noise_type == "uniform":
                frac = torch.rand(hist.shape[0], generator=rng, device=hist.device) * 2.0 * self.noise_scale
            elif self.noise_type == "gaussian":
                frac = torch.randn(hist.shape[0], generator=rng, device=hist.device) * self.noise_scale
            else:
                frac = torch.zeros(hist.shape, device=hist.device)
            hist = torch.clamp(hist * (1.0 + frac), 0.0, None)
        return hist

    # rows of a (d x d) transport matrix consumed by this diagnostic: list of [d] vectors
    def projection_rows(self, matrix: torch.Tensor) -> List[torch.Tensor]:
        raise NotImplementedError

    # A histogram is a SUM over particles followed by a normalisation: raw_sums(x, rows) are the per-projection sums of the
    # particles at hand (kernel sums for kde=True, bin counts otherwise), from_sums(S, n) finishes [P, bins...] histograms from
    # sums over n particles.  batched = from_sums(raw_sums); a data-parallel run all-reduces the sums in between
    # (MENTFlow.loss, simulate.raw_sums).
    def raw_sums(self, x: torch.Tensor, rows: List[torch.Tensor]) -> torch.Tensor:
        raise NotImplementedError

    def from_sums(self, S: torch.Tensor, n_total: int) -> torch.Tensor:
        raise NotImplementedError

    def identity_rows(self, x: torch.Tensor) -> List[torch.Tensor]:
        eye = torch.eye(x.shape[1], dtype=x.dtype, device=x.device)
        return [r[None, :] for r in self.projection_rows(eye)]

    def batched(self, x: torch.Tensor, rows: List[torch.Tensor]) -> torch.Tensor:
        """All P projections in one launch: [P, bins...] normalised histograms (kde) or densities (hard bins)."""
        return self.from_sums(self.raw_sums(x, rows), x.shape[0])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._apply_noise(self.batched(x, self.identity_rows(x))[0])


class Histogram1D(Histogram):
    """diagnostics.py:71-131."""

    def __init__(self, edges: torch.Tensor, bandwidth: Optional[float] = None, axis: int = 0,
                 direction: torch.Tensor = None, kde: bool = True, **kws) -> None:
        super().__init__(**kws)
        self.axis = axis
        self.kde = kde
        self.ndim = 1
        self.direction = direction
        if self.direction is not None:
            self.direction = self.direction / torch.norm(self.direction)
        if bandwidth is None:
            bandwidth = 0.5
        self.bandwidth_bins = float(bandwidth)
        self.register_buffer("edges", edges)
        self.register_buffer("coords", coords_from_edges(self.edges))
        self.register_buffer("resolution", edges[1] - edges[0])
        self.register_buffer("bandwidth", bandwidth * self.resolution)
        # host copies of the two scalars every launch needs (read once here: no device->host sync per call, and the
        # step stays capturable into a hipGraph)
        self.resolution_value = float(self.resolution)
        self.bandwidth_value = float(self.bandwidth)

    def projection_rows(self, matrix: torch.Tensor) -> List[torch.Tensor]:
        """u[:, axis] = x . matrix[axis]  (or  (x @ M.T) . direction = x . (direction @ M))."""
        if self.direction is None:
            return [matrix[self.axis]]
        return [self.direction.to(matrix) @ matrix]

    def raw_sums(self, x: torch.Tensor, rows: List[torch.Tensor]) -> torch.Tensor:
        V = rows[0].to(torch.float32).contiguous()
        if self.kde:
            return ops.ProjKde1dFn.apply(x, V, self.coords, self.bandwidth_value, ops.kde_radius(self.bandwidth_bins))
        return ops.proj_hist_counts_1d(x.detach(), V, self.edges).to(torch.float32)

    def from_sums(self, S: torch.Tensor, n_total: int) -> torch.Tensor:
        if self.kde:
            ghat, _ = ops.HistNormDiscFn.apply(S, None, True, 1.0 / n_total, self.resolution_value, 1.0e-10, 0, 0.0, 1.0)
            return ghat
        widths = (self.edges[1:] - self.edges[:-1])[None, :]
        return S / S.sum(dim=1, keepdim=True) / widths                    # torch.histogram(density=True)


class Histogram2D(Histogram):
    """diagnostics.py:134-201."""

    def __init__(self, axis: Iterable[int], edges: Iterable[torch.Tensor], bandwidth: Iterable[Optional[float]] = (None, None),
                 kde: bool = True, **kws) -> None:
        super().__init__(**kws)
        self.axis = tuple(axis)
        self.kde = kde
        self.ndim = 2
        bx, by = bandwidth
        bx = 0.5 if bx is None else bx
        by = 0.5 if by is None else by
        self.bandwidth_bins = (float(bx), float(by))
        self.register_buffer("edges_x", edges[0])
        self.register_buffer("edges_y", edges[1])
        self.register_buffer("coords_x", coords_from_edges(self.edges_x))
        self.register_buffer("coords_y", coords_from_edges(self.edges_y))
        self.register_buffer("resolution_x", self.edges_x[1] - self.edges_x[0])
        self.register_buffer("resolution_y", self.edges_y[1] - self.edges_y[0])
        self.register_buffer("bandwidth_x", bx * self.resolution_x)
        self.register_buffer("bandwidth_y", by * self.resolution_y)
        self.resolution_values = (float(self.resolution_x), float(self.resolution_y))
        self.bandwidth_values = (float(self.bandwidth_x), float(self.bandwidth_y))

    @property
    def edges(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return (self.edges_x, self.edges_y)

    def projection_rows(self, matrix: torch.Tensor) -> List[torch.Tensor]:
        return [matrix[self.axis[0]], matrix[self.axis[1]]]

    def raw_sums(self, x: torch.Tensor, rows: List[torch.Tensor]) -> torch.Tensor:
        V0 = rows[0].to(torch.float32).contiguous()
        V1 = rows[1].to(torch.float32).contiguous()
        if self.kde:
            return ops.ProjKde2dFn.apply(x, V0, V1, self.coords_x, self.coords_y, self.bandwidth_values[0],
                                         self.bandwidth_values[1], ops.kde_radius(self.bandwidth_bins[0]),
                                         ops.kde_radius(self.bandwidth_bins[1]))
        return ops.proj_hist_counts_2d(x.detach(), V0, V1, self.edges_x, self.edges_y).to(torch.float32)

    def from_sums(self, S: torch.Tensor, n_total: int) -> torch.Tensor:
        if self.kde:
            P, Bx, By = S.shape
            ghat, _ = ops.HistNormDiscFn.apply(S.reshape(P, Bx * By), None, True, 1.0,
                                               self.resolution_values[0] * self.resolution_values[1], 1.0e-10, 0, 0.0, 1.0)
            return ghat.view(P, Bx, By)
        area = (self.edges_x[1:] - self.edges_x[:-1])[:, None] * (self.edges_y[1:] - self.edges_y[:-1])[None, :]
        return S / S.sum(dim=(1, 2), keepdim=True) / area[None]             # np.histogramdd(density=True)


class Projection(Diagnostic):
    """Projects points onto axis (no density estimation) — diagnostics.py:204-211."""

    def __init__(self, axis: Union[int, Tuple[int]], **kws) -> None:
        super().__init__(**kws)
        self.axis = axis

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x[:, self.axis]
