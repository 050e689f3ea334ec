"""Synthetic ground-truth samplers used by the benchmark configs (host side, numpy RNG; runs once per experiment).

Restates the three distributions the BASELINE configs name — mentflow/distributions/distributions.py:
Rings :244-273 (built on KV :181-191), GaussianMixture :136-160, SwissRoll :276-286 — together with the
post-processing of ``Distribution.sample_np`` (:46-61: shuffle, optional noise) and utils.py (corrupt, shuffle).
Pinned against samples of the reference itself in tests/golden/ref_dist_*.npz."""
from __future__ import annotations

import numpy as np
import scipy.special
import torch


def _sphere_surface_area(r=1.0, d=3):
    """mentflow/utils/utils.py:35-38."""
    factor = 2.0 * np.pi ** (0.5 * d) / scipy.special.gamma(0.5 * d)
    return factor * (r ** (d - 1))


def _unit_std(a: np.ndarray) -> np.ndarray:
    """Every column scaled to unit standard deviation, in place for float arrays (the reference's `x /= np.std(x, 0)`)."""
    a /= a.std(axis=0)
    return a


class Distribution:
    """Seeded sampler: `_sample` draws the raw cloud, `sample_np` adds the reference's post-processing (one permutation,
    then optional Gaussian noise — in this order, from the same generator: the golden fixtures pin the stream)."""

    default_noise = None

    def __init__(self, ndim: int = 2, seed: int = None, noise: float = None, shuffle: bool = True) -> None:
        self.ndim, self.seed, self.shuffle = ndim, seed, shuffle
        self.noise = noise if noise is not None else self.default_noise
        self.rng = np.random.default_rng(seed)

    def _sample(self, size: int) -> np.ndarray:
        raise NotImplementedError(type(self).__name__)

    def sample_np(self, size: int) -> np.ndarray:
        cloud = self._sample(int(size))
        if self.shuffle:
            cloud = self.rng.permutation(cloud)
        if self.noise:
            cloud = cloud + self.rng.normal(scale=self.noise, size=cloud.shape)
        return cloud

    def sample(self, size: int) -> torch.Tensor:
        return torch.from_numpy(self.sample_np(size)).to(torch.float32)


class Gaussian(Distribution):
    def _sample(self, size):
        return self.rng.normal(size=(size, self.ndim))


class KV(Distribution):
    """Uniform on the unit sphere surface, columns rescaled to unit standard deviation."""

    default_noise = 0.05

    def _sample(self, size):
        g = self.rng.normal(size=(size, self.ndim))
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        return _unit_std(g)


class Rings(Distribution):
    """Concentric KV shells; populations ~ surface area x a linear decay factor (outer shell = `decay`)."""

    default_noise = 0.15

    def __init__(self, n_rings: int = 2, decay: float = 0.5, **kws) -> None:
        super().__init__(**kws)
        self.n_rings, self.decay = n_rings, decay

    def _sample(self, size):
        radii = np.linspace(1.0, 0.0, self.n_rings, endpoint=False)[::-1]
        weights = np.array([_sphere_surface_area(r=r, d=self.ndim) for r in radii]) * np.linspace(1.0, self.decay, self.n_rings)
        counts = (weights * (size / weights.sum())).astype(int)
        shell = KV(ndim=self.ndim, seed=self.seed)          # ONE sampler for all shells: its stream continues
        parts = [(r * shell.sample(c)).numpy() for c, r in zip(counts, radii)]        # float32, as the reference
        return _unit_std(np.vstack(parts))


class GaussianMixture(Distribution):
    """`modes` isotropic Gaussians with centres drawn uniformly in [-xmax, xmax]^d at construction."""

    def __init__(self, modes: int = 7, xmax: float = 3.0, scale: float = 0.75, shiftscale=True, **kws) -> None:
        super().__init__(**kws)
        self.modes, self.shiftscale = modes, shiftscale
        self.locs = self.rng.uniform(-xmax, xmax, size=(modes, self.ndim))
        self.scales = np.full(modes, scale, dtype=np.float64)

    def _sample(self, size):
        per_mode = size // self.modes
        cloud = np.vstack([self.rng.normal(loc=centre, scale=width, size=(per_mode, self.ndim))
                           for width, centre in zip(self.scales, self.locs)])
        if self.shiftscale:
            cloud = cloud - cloud.mean(axis=0)
            cloud = cloud / cloud.std(axis=0)
        return cloud


class SwissRoll(Distribution):
    default_noise = 0.15

    def _sample(self, size):
        t = 1.5 * np.pi * (1.0 + 2.0 * self.rng.uniform(0.0, 1.0, size=size))
        return _unit_std(np.stack([t * np.cos(t), t * np.sin(t)], axis=-1))


DISTRIBUTIONS = {"gaussian": Gaussian, "kv": KV, "rings": Rings, "gaussian_mixture": GaussianMixture,
                 "swissroll": SwissRoll}


def get_distribution(name: str, **kws) -> Distribution:
    try:
        cls = DISTRIBUTIONS[name]
    except KeyError:
        raise NotImplementedError(f"distribution '{name}' is not restated (built: {sorted(DISTRIBUTIONS)})") from None
    return cls(**kws)
