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
from .transform import LinearTransform


def group_measurements(transforms, diagnostics) -> Dict[int, Tuple[Histogram, List[Tuple[int, int]], List[List[torch.Tensor]]]]:
    """{id(diagnostic): (diagnostic, [(i, j) slots], [rows_k stacked later])} for the fused path; raises for
    anything the fused kernels do not cover (nothing silently falls back to eager torch)."""
    groups: Dict[int, Tuple[Histogram, List[Tuple[int, int]], List[List[torch.Tensor]]]] = {}
    for i, transform in enumerate(transforms):
        if not isinstance(transform, LinearTransform):
            raise NotImplementedError(
                f"{type(transform).__name__}: only LinearTransform is on the MI355X hot path (SURVEY.md §8a a7)")
        for j, diagnostic in enumerate(diagnostics[i]):
            if not isinstance(diagnostic, Histogram):
                raise NotImplementedError(
                    f"{type(diagnostic).__name__}: only Histogram1D/Histogram2D diagnostics are on the hot path")
            entry = groups.setdefault(id(diagnostic), (diagnostic, [], []))
            entry[1].append((i, j))
            entry[2].append(diagnostic.projection_rows(transform.matrix))
    return groups


def forward(x: torch.Tensor, transforms: List[nn.Module], diagnostics: List[List[nn.Module]]) -> List[List[torch.Tensor]]:
    """simulate.py:8-33: predictions[i][j] = diagnostics[i][j](transforms[i](x))."""
    predictions: List[List[torch.Tensor]] = [[None] * len(diagnostics[i]) for i in range(len(transforms))]
    for diagnostic, slots, rows in group_measurements(transforms, diagnostics).values():
        stacked = [torch.stack([r[k] for r in rows]) for k in range(len(rows[0]))]
        hists = diagnostic.batched(x, stacked)
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
