"""Hydra-free experiment harness: plain-dict/kwargs version of the reference's
``generate_training_data -> setup_mentflow_model`` (experiments/setup.py:21-151) and of the per-experiment
``make_transforms / make_diagnostics`` (experiments/rec_nd_1d/setup.py:28-69, rec_2d/linear/setup.py:27-60,
rec_nd_2d/setup.py:34-90), with the same key names (ndim, seed, meas.{num,xmax,bins,bandwidth}, gen.*, model.*).
hydra/omegaconf are not available on the GPU box; the hot-path objects built here are the same ones the reference's
scripts would build."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import diagnostics as mfd
from . import entropy as mfe
from . import loss as mfl
from . import prior as mfp
from . import simulate as mfs
from .core import MENTFlow
from .distributions import get_distribution
from .generate import build_generator
from .utils import unravel


def make_directions(num: int, ndim: int, seed: Optional[int]) -> torch.Tensor:
    """rec_nd_1d/setup.py:28-37 ("isotropic" optics), CPU generator stream (the reference draws on cfg.device)."""
    rng = torch.Generator(device="cpu")
    if seed is not None:
        rng.manual_seed(seed)
    directions = torch.randn((num, ndim), generator=rng, device="cpu")
    return directions / torch.norm(directions, dim=1)[:, None]


def make_transforms_nd_1d(num: int, ndim: int, seed: Optional[int]) -> List[mfs.LinearTransform]:
    """rec_nd_1d/setup.py:39-49."""
    transforms = []
    for direction in make_directions(num, ndim, seed):
        M = torch.eye(ndim)
        M[0, :] = direction
        transforms.append(mfs.LinearTransform(M.float()))
    return transforms


def make_transforms_2d_linear(num: int, min_angle: float = 0.0, max_angle: float = 180.0):
    """rec_2d/linear/setup.py:27-43."""
    angles = np.linspace(np.radians(min_angle), np.radians(max_angle), num, endpoint=False)
    return [mfs.LinearTransform(mfs.rotation_matrix(a).type(torch.float32)) for a in angles]


def make_transforms_nd_2d_corner(ndim: int) -> List[mfs.LinearTransform]:
    """rec_nd_2d/setup.py:38-53."""
    transforms = []
    for i in range(ndim):
        for j in range(i):
            matrices = []
            for k, l in zip((0, 2), (j, i)):
                matrix = torch.eye(ndim)
                matrix[k, k] = matrix[l, l] = 0.0
                matrix[k, l] = matrix[l, k] = 1.0
                matrices.append(matrix.float())
            transforms.append(mfs.LinearTransform(torch.linalg.multi_dot(matrices[::-1])))
    return transforms


def make_transforms_nd_2d_random(num: int, ndim: int, seed: Optional[int]) -> List[mfs.LinearTransform]:
    """BASELINE config C5 (synthetic extension of the 15 'corner' views to `num` views): two
    orthonormal random directions written into rows 0 and 2 of an identity LinearTransform."""
    rng = torch.Generator(device="cpu")
    if seed is not None:
        rng.manual_seed(seed)
    transforms = []
    for _ in range(num):
        a = torch.randn(ndim, generator=rng)
        b = torch.randn(ndim, generator=rng)
        a = a / a.norm()
        b = b - (a @ b) * a
        b = b / b.norm()
        M = torch.eye(ndim)
        M[0, :] = a
        M[2, :] = b
        transforms.append(mfs.LinearTransform(M.float()))
    return transforms


def generate_training_data(x_true: torch.Tensor, transforms, diagnostic, device):
    """experiments/setup.py:31-75: measurements = hard-binned (kde=False) histograms of the ground truth, then
    renormalised; the SAME diagnostic object is shared by every transform (:43-44)."""
    with torch.no_grad():
        transforms = [t.to(device) for t in transforms]
        diagnostic = diagnostic.to(device)
        diagnostics = [[diagnostic] for _ in transforms]
        x_true = x_true.type(torch.float32).to(device)
        for d in unravel(diagnostics):
            d.kde, d.noise = False, True
        measurements = mfs.forward(x_true, transforms, diagnostics)
        for d in unravel(diagnostics):
            d.kde, d.noise = True, False
        for i in range(len(measurements)):
            for j in range(len(measurements[i])):
                m, d = measurements[i][j], diagnostics[i][j]
                if m.ndim == 1:
                    volume = d.edges[1] - d.edges[0]
                else:
                    volume = math.prod([e[1] - e[0] for e in d.edges])
                measurements[i][j] = m / torch.sum(m) / volume
    return transforms, diagnostics, measurements


@dataclass
class Problem:
    model: MENTFlow
    transforms: list
    diagnostics: list
    measurements: list
    cfg: dict = field(default_factory=dict)


def make_transforms_2d_nonlinear(num: int = 4, max_angle: float = 90.0, order: int = 3, max_strength: float = 1.5):
    """rec_2d/nonlinear/setup.py:24-44: constant rotation, multipole strength swept over [-max, +max]."""
    out = []
    for strength in np.linspace(-max_strength, max_strength, num):
        rotation = mfs.LinearTransform(mfs.rotation_matrix(np.radians(max_angle)).type(torch.float32))
        out.append(mfs.CompositeTransform(mfs.MultipoleTransform(order=order, strength=float(strength)), rotation))
    return out


def build_problem(ndim: int = 6, num: int = 25, bins: int = 64, xmax: float = 4.0, seed: Optional[int] = 2,
                  transforms: int = 5, prior_scale: float = 1.0, device="cuda", dist_name: str = "rings",
                  dist_kws: Optional[dict] = None, meas_samples: int = 1_000_000, gen_name: str = "nsf",
                  hidden_layers: int = 3, hidden_units: int = 64, spline_bins: int = 20, optics: str = "nd_1d",
                  bandwidth: float = 0.5, discrepancy: str = "kld", penalty_parameter: float = 0.0) -> Problem:
    """generate_training_data + setup_mentflow_model (experiments/setup.py:21-151) for the BASELINE workloads."""
    device = torch.device(device)
    if seed is not None:
        torch.manual_seed(seed)                                           # setup.py:35-36
    if optics == "nd_1d":
        tfs = make_transforms_nd_1d(num, ndim, seed)
        diag = mfd.Histogram1D(axis=0, edges=torch.linspace(-xmax, xmax, bins + 1), bandwidth=bandwidth, noise=True,
                               noise_scale=0.0, device=device, seed=seed)
    elif optics == "2d_linear":
        tfs = make_transforms_2d_linear(num)
        diag = mfd.Histogram1D(axis=0, edges=torch.linspace(-xmax, xmax, bins + 1), bandwidth=bandwidth, noise=True,
                               noise_scale=0.0, device=device, seed=seed)
    elif optics == "2d_nonlinear":
        tfs = make_transforms_2d_nonlinear(num)
        diag = mfd.Histogram1D(axis=0, edges=torch.linspace(-xmax, xmax, bins + 1), bandwidth=bandwidth, noise=True,
                               noise_scale=0.0, device=device, seed=seed)
    elif optics in ("nd_2d_corner", "nd_2d_random"):
        tfs = make_transforms_nd_2d_corner(ndim) if optics == "nd_2d_corner" else make_transforms_nd_2d_random(num, ndim, seed)
        e = [torch.linspace(-xmax, xmax, bins + 1), torch.linspace(-xmax, xmax, bins + 1)]
        diag = mfd.Histogram2D(axis=(0, 2), edges=e, bandwidth=(bandwidth, bandwidth), noise=True, noise_scale=0.0,
                               device=device, seed=seed)
    else:
        raise ValueError(f"Invalid optics '{optics}'")
    kws = dict(dist_kws or {})
    if dist_name == "rings":
        kws.setdefault("decay", 0.2)                                      # run_rings.sh:40
    x_true = get_distribution(dist_name, ndim=ndim, seed=seed, **kws).sample(meas_samples)
    tfs, diagnostics, measurements = generate_training_data(x_true, tfs, diag, device)

    gen_kws = dict(input_features=ndim, output_features=ndim, hidden_layers=hidden_layers, hidden_units=hidden_units)
    if gen_name == "nn":
        gen_kws.update(dropout=0.0, activation="tanh")                    # config/gen/nn.yaml
    else:
        gen_kws["transforms"] = transforms
    if gen_name == "nsf":
        gen_kws["bins"] = spline_bins                                     # setup.py:120-121
    generator = build_generator(gen_name, device=device, **gen_kws)
    prior = mfp.Gaussian(ndim=ndim, scale=prior_scale, device=device)     # setup.py:133-135
    disc = {"kld": mfl.kl_divergence, "mae": mfl.mean_absolute_error, "mse": mfl.mean_square_error}[discrepancy]
    # config/model/nn.yaml: entropy_estimator "none" for the density-free NN generator (setup.py:137-141)
    estimator = mfe.EmptyEntropyEstimator() if gen_name == "nn" else mfe.MonteCarloEntropyEstimator(prior=prior)
    model = MENTFlow(generator=generator, entropy_estimator=estimator, prior=prior,
                     transforms=tfs, diagnostics=diagnostics, measurements=measurements,
                     penalty_parameter=penalty_parameter, discrepancy_function=disc)
    model = model.to(device)
    cfg = dict(ndim=ndim, num=num, bins=bins, xmax=xmax, seed=seed, transforms=transforms, prior_scale=prior_scale,
               gen_name=gen_name, hidden_layers=hidden_layers, spline_bins=spline_bins, optics=optics,
               bandwidth=bandwidth, discrepancy=discrepancy)
    return Problem(model, tfs, diagnostics, measurements, cfg)
