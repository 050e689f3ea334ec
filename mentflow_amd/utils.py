"""Small host helpers (names of mentflow/utils/utils.py:10-15 and mentflow/utils/grid.py:5-10)."""
from itertools import chain
from typing import Iterable, Iterator

import numpy as np
import torch


def unravel(nested: Iterable[Iterable]) -> Iterator:
    """Flatten one level: [[a, b], [c]] -> a, b, c  (measurements / diagnostics are lists of lists)."""
    return chain.from_iterable(nested)


def grab(t: torch.Tensor) -> np.ndarray:
    """Detached host copy as a numpy array."""
    return t.detach().cpu().numpy()


def coords_from_edges(edges: torch.Tensor) -> torch.Tensor:
    """Bin centres of a 1-D edge vector."""
    return (edges[1:] + edges[:-1]) * 0.5


def get_grid_points(*coords: torch.Tensor) -> torch.Tensor:
    """[prod(len(c)), len(coords)] points of the tensor-product grid, first coordinate slowest ("ij" order)."""
    mesh = torch.meshgrid(*coords, indexing="ij")
    return torch.stack([m.reshape(-1) for m in mesh], dim=1)
