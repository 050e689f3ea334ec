"""Small helpers — mirrors mentflow/utils/utils.py:10-15 and mentflow/utils/grid.py:5-10."""
import itertools

import torch


def unravel(iterable):
    return itertools.chain.from_iterable(iterable)


def grab(x):
    return x.detach().cpu().numpy()


def coords_from_edges(edges: torch.Tensor) -> torch.Tensor:
    return 0.5 * (edges[:-1] + edges[1:])


def get_grid_points(*coords: torch.Tensor) -> torch.Tensor:
    return torch.vstack([C.ravel() for C in torch.meshgrid(*coords, indexing="ij")]).T
