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


class Distribution:
    default_noise = None

    def __init__(self, ndim: int = 2, seed: int = None, noise: float = None, shuffle: bool = True) -> None:
        self.ndim, self.seed, self.shuffle = ndim, seed, shuffle
        self.rng = np.random.default_rng(seed)
        self.noise = self.default_noise if noise is None else noise

    def _sample(self, size: int) -> np.ndarray:
        raise NotImplementedError

    def sample_np(self, size: int) -> np.ndarray:
        x = self._sample(int(size))
        if self.shuffle:
            x = self.rng.permutation(x)
        if self.noise:
            x = x + self.rng.normal(scale=self.noise, size=x.shape)
        return x

    def sample(self, size: int) -> torch.Tensor:
        return torch.from_numpy(self.sample_np(size)).type(torch.float32)


class Gaussian(Distribution):
    def _sample(self, size):
        return self.rng.normal(size=(size, self.ndim))


class KV(Distribution):
    default_noise = 0.05

    def _sample(self, size):
        x = self.rng.normal(size=(size, self.ndim))
        x /= np.linalg.norm(x, axis=1)[:, None]
        x /= np.std(x, axis=0)
        return x


class Rings(Distribution):
    default_noise = 0.15

    def __init__(self, n_rings: int = 2, decay: float = 0.5, **kws) -> None:
        super().__init__(**kws)
        self.n_rings, self.decay = n_rings, decay

    def _sample(self, size):
        radii = np.linspace(1.0, 0.0, self.n_rings, endpoint=False)[::-1]
        sizes = np.array([_sphere_surface_area(d=self.ndim, r=r) for r in radii])
        sizes = sizes * np.linspace(1.0, self.decay, self.n_rings)
        sizes = (sizes * (size / np.sum(sizes))).astype(int)
        dist = KV(ndim=self.ndim, seed=self.seed)
        x = [(radius * dist.sample(s)).numpy() for s, radius in zip(sizes, radii)]   # float32, as the reference
        x = np.vstack(x)
        x /= np.std(x, axis=0)
        return x


class GaussianMixture(Distribution):
    def __init__(self, modes: int = 7, xmax: float = 3.0, scale: float = 0.75, shiftscale=True, **kws) -> None:
        super().__init__(**kws)
        self.modes = modes
        self.locs = self.rng.uniform(-xmax, xmax, size=(self.modes, self.ndim))
        self.scales = scale * np.ones(self.modes)
        self.shiftscale = shiftscale

    def _sample(self, size):
        x = np.vstack([self.rng.normal(loc=loc, scale=scale, size=(size // self.modes, self.ndim))
                       for scale, loc in zip(self.scales, self.locs)])
        if self.shiftscale:
            x = x - np.mean(x, axis=0)
            x = x / np.std(x, axis=0)
        return x


class SwissRoll(Distribution):
    default_noise = 0.15

    def _sample(self, size):
        t = 1.5 * np.pi * (1.0 + 2.0 * self.rng.uniform(0.0, 1.0, size=size))
        x = np.stack([t * np.cos(t), t * np.sin(t)], axis=-1)
        x /= np.std(x, axis=0)
        return x


DISTRIBUTIONS = {"gaussian": Gaussian, "kv": KV, "rings": Rings, "gaussian_mixture": GaussianMixture,
                 "swissroll": SwissRoll}


def get_distribution(name: str, **kws) -> Distribution:
    if name not in DISTRIBUTIONS:
        raise NotImplementedError(f"distribution '{name}' is not restated (built: {sorted(DISTRIBUTIONS)})")
    return DISTRIBUTIONS[name](**kws)
