"""Forward simulation of the measurement set — mirrors mentflow/simulate/simulate.py:8-47.

The reference loops over transforms in Python (``u = transform(x.clone())`` then each diagnostic).  Here every
(transform, diagnostic) pair of a measurement set that is ``LinearTransform`` + ``Histogram1D/2D`` is reduced to its
projection row(s) ``matrix[axis]`` and all pairs that share a diagnostic object are evaluated by ONE fused
projection + KDE launch; the list-of-lists result has the same structure and values as the reference's.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from ..diagnostics import Histogram
from .transform import CompositeTransform, LinearTransform


def split_transform(transform):
    """(pre, linear): `linear` is the LinearTransform whose rows the fused projection kernels consume; `pre` is the chain
    of transforms applied before it (empty for a plain LinearTransform; e.g. a MultipoleTransform kick for the
    CompositeTransform(multipole, rotation) of experiments/rec_2d/nonlinear/setup.py:35-43)."""
    if isinstance(transform, LinearTransform):
        return (), transform
    if isinstance(transform, CompositeTransform):
        chain = list(transform.transforms)
        if chain and isinstance(chain[-1], LinearTransform):
            return tuple(chain[:-1]), chain[-1]
    raise NotImplementedError(
        f"{type(transform).__name__}: the fused projection path needs a LinearTransform (optionally behind a "
        "CompositeTransform of pre-transforms); SURVEY.md §8a a7")


def group_measurements(transforms, diagnostics):
    """{(id(diagnostic), pre chain): (diagnostic, pre, [(i, j) slots], [projection rows])} for the fused path; raises for
    anything the fused kernels do not cover (nothing silently falls back to eager torch).  All (transform, diagnostic)
    pairs that share a diagnostic object and the same pre-transform chain are evaluated by one kernel launch."""
    groups = {}
    for i, transform in enumerate(transforms):
        pre, linear = split_transform(transform)
        for j, diagnostic in enumerate(diagnostics[i]):
            if not isinstance(diagnostic, Histogram):
                raise NotImplementedError(
                    f"{type(diagnostic).__name__}: only Histogram1D/Histogram2D diagnostics are on the hot path")
            key = (id(diagnostic), tuple(id(t) for t in pre))
            entry = groups.setdefault(key, (diagnostic, pre, [], []))
            entry[2].append((i, j))
            entry[3].append(diagnostic.projection_rows(linear.matrix))
    return groups


def apply_pre(x: torch.Tensor, pre) -> torch.Tensor:
    for t in pre:
        x = t(x)
    return x


def forward(x: torch.Tensor, transforms: List[nn.Module], diagnostics: List[List[nn.Module]]) -> List[List[torch.Tensor]]:
    """simulate.py:8-33: predictions[i][j] = diagnostics[i][j](transforms[i](x))."""
    predictions: List[List[torch.Tensor]] = [[None] * len(diagnostics[i]) for i in range(len(transforms))]
    for diagnostic, pre, slots, rows in group_measurements(transforms, diagnostics).values():
        stacked = [torch.stack([r[k] for r in rows]) for k in range(len(rows[0]))]
        hists = diagnostic.batched(apply_pre(x, pre), stacked)
        for (i, j), h in zip(slots, hists.unbind(0)):
            predictions[i][j] = diagnostic._apply_noise(h)
    return predictions


class Simulator:
    """simulate.py:36-47."""

    def __init__(self, transforms, diagnostics) -> None:
        self.transforms = transforms
        self.diagnostics = diagnostics

    def forward(self, x: torch.Tensor) -> List[List[torch.Tensor]]:
        return forward(x, self.transforms, self.diagnostics)
