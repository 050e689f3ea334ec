"""Forward simulation of the measurement set — mirrors mentflow/simulate/simulate.py:8-47.

The reference loops over transforms in Python (``u = transform(x.clone())`` then each diagnostic).  Here every
(transform, diagnostic) pair of a measurement set that is ``LinearTransform`` (optionally behind a ``CompositeTransform`` of
pre-transforms) + ``Histogram1D/2D`` is reduced to its projection row(s) ``matrix[axis]`` and all pairs that share a
diagnostic object are evaluated by ONE fused projection + KDE launch (the fast path).  Any other pair — a transport that is
not of that form (e.g. a kick applied AFTER the rotation), any other ``nn.Module`` transport, a ``Projection`` or user
diagnostic — takes the reference's generic loop: the transform is applied as given and the diagnostic is called on the
result (``Histogram*.forward(u)`` still runs the HIP kernels, with identity projection rows).  The list-of-lists result has
the same structure and values as the reference's either way; which path a measurement set takes is logged once.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from ..diagnostics import Histogram
from .transform import CompositeTransform, LinearTransform

log = logging.getLogger("mentflow_amd.simulate")
_logged_plans = set()


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
    """(groups, generic): groups = {(id(diagnostic), pre chain): (diagnostic, pre, [(i, j) slots], [projection rows])} for
    the fused path — all (transform, diagnostic) pairs that share a diagnostic object and the same pre-transform chain are
    evaluated by one kernel launch; generic = the (i, j) slots the fused kernels do not cover (simulate.py:30-33 loop)."""
    groups, generic = {}, []
    for i, transform in enumerate(transforms):
        try:
            pre, linear = split_transform(transform)
        except NotImplementedError:
            pre, linear = (), None
        for j, diagnostic in enumerate(diagnostics[i]):
            if linear is None or not isinstance(diagnostic, Histogram):
                generic.append((i, j))
                continue
            key = (id(diagnostic), tuple(id(t) for t in pre))
            entry = groups.setdefault(key, (diagnostic, pre, [], []))
            entry[2].append((i, j))
            entry[3].append(diagnostic.projection_rows(linear.matrix))
    return groups, generic


def apply_pre(x: torch.Tensor, pre) -> torch.Tensor:
    for t in pre:
        x = t(x)
    return x


def _log_plan(transforms, diagnostics, groups, generic) -> None:
    key = (tuple(id(t) for t in transforms), tuple(id(d) for row in diagnostics for d in row))
    if key in _logged_plans:
        return
    _logged_plans.add(key)
    n_fused = sum(len(g[2]) for g in groups.values())
    log.info("simulate.forward: %d measurement(s) on the fused projection + KDE path (%d launch group(s)), %d on the "
             "generic transform -> diagnostic loop%s", n_fused, len(groups), len(generic),
             "" if not generic else " " + str([(type(transforms[i]).__name__, type(diagnostics[i][j]).__name__)
                                               for i, j in generic[:4]]))


def raw_sums(x: torch.Tensor, transforms: List[nn.Module], diagnostics: List[List[nn.Module]]):
    """The measurement set as SUMS over the particles at hand, for data-parallel runs: (pieces, finish) where `pieces` are the
    raw per-projection sums of every (transform, diagnostic) pair — fused groups and generic pairs alike — and
    finish(reduced_pieces, n_total) turns the (all-reduced) sums into the reference's predictions[i][j] list
    (simulate.py:8-33).  Needs diagnostics with a sum form (Histogram1D / Histogram2D); anything else raises, naming it."""
    groups, generic = group_measurements(transforms, diagnostics)
    _log_plan(transforms, diagnostics, groups, generic)
    pieces, recipe = [], []
    for diagnostic, pre, slots, rows in groups.values():
        stacked = [torch.stack([r[k] for r in rows]) for k in range(len(rows[0]))]
        pieces.append(diagnostic.raw_sums(apply_pre(x, pre), stacked))
        recipe.append((diagnostic, slots))
    transported: Dict[int, torch.Tensor] = {}
    for i, j in generic:
        diagnostic = diagnostics[i][j]
        if not isinstance(diagnostic, Histogram):
            raise NotImplementedError(
                f"data-parallel simulation: {type(diagnostic).__name__} (measurement [{i}][{j}]) has no sum form to reduce over "
                "the ranks; histogram diagnostics do (Histogram1D / Histogram2D)")
        if i not in transported:
            transported[i] = transforms[i](x.clone())
        pieces.append(diagnostic.raw_sums(transported[i], diagnostic.identity_rows(transported[i])))
        recipe.append((diagnostic, [(i, j)]))

    def finish(reduced, n_total: int) -> List[List[torch.Tensor]]:
        predictions: List[List[torch.Tensor]] = [[None] * len(diagnostics[i]) for i in range(len(transforms))]
        for (diagnostic, slots), S in zip(recipe, reduced):
            for (i, j), h in zip(slots, diagnostic.from_sums(S, n_total).unbind(0)):
                predictions[i][j] = diagnostic._apply_noise(h)
        return predictions

    return pieces, finish


def forward(x: torch.Tensor, transforms: List[nn.Module], diagnostics: List[List[nn.Module]]) -> List[List[torch.Tensor]]:
    """simulate.py:8-33: predictions[i][j] = diagnostics[i][j](transforms[i](x))."""
    predictions: List[List[torch.Tensor]] = [[None] * len(diagnostics[i]) for i in range(len(transforms))]
    groups, generic = group_measurements(transforms, diagnostics)
    _log_plan(transforms, diagnostics, groups, generic)
    for diagnostic, pre, slots, rows in groups.values():
        stacked = [torch.stack([r[k] for r in rows]) for k in range(len(rows[0]))]
        hists = diagnostic.batched(apply_pre(x, pre), stacked)
        for (i, j), h in zip(slots, hists.unbind(0)):
            predictions[i][j] = diagnostic._apply_noise(h)
    transported: Dict[int, torch.Tensor] = {}
    for i, j in generic:                                  # the reference's loop, one transform application per i
        if i not in transported:
            transported[i] = transforms[i](x.clone())     # simulate.py:32 (a user transform may work in place)
        predictions[i][j] = diagnostics[i][j](transported[i])
    return predictions


class Simulator:
    """simulate.py:36-47."""

    def __init__(self, transforms, diagnostics) -> None:
        self.transforms = transforms
        self.diagnostics = diagnostics

    def forward(self, x: torch.Tensor) -> List[List[torch.Tensor]]:
        return forward(x, self.transforms, self.diagnostics)
