"""Autoregressive masks of the conditioner, in closed form.

Reference behaviour: zuko 1.3.1 ``MaskedAutoregressiveTransform.__init__`` + ``nn.MaskedMLP.__init__`` as used by
``mentflow/generate/build.py:24-40`` (``passes = features``: fully autoregressive).  zuko derives the masks from the
unique rows of the adjacency matrix and a precedence relation; for the adjacency ``order[i] > order[j]`` that procedure
reduces to *dependency classes*: output feature i may see inputs of order < order[i]; hidden unit u of every hidden
layer gets class ``c_u = 1 + (u mod (d-1))`` (zuko cycles the hidden units over the d-1 non-empty dependency sets);

    input  -> hidden : mask[u, j]  = order[j] <  c_u
    hidden -> hidden : mask[u, u'] = c_u'     <= c_u
    hidden -> output : mask[(i, t), u] = c_u  <= order[i]      (t = 0..total-1 parameters of feature i)

The feature of order 0 therefore has an all-zero output mask (its parameters are pure bias).
tests/test_host_logic.py checks these against the literal zuko procedure restated in oracle/flow.py.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def hidden_classes(features: int, width: int) -> torch.Tensor:
    if features < 2:
        raise ValueError("The adjacency matrix leads to a null Jacobian.")   # zuko's error for d = 1
    return 1 + (torch.arange(width) % (features - 1))


def conditioner_masks(order: torch.Tensor, hidden_features: Sequence[int], total: int) -> List[torch.Tensor]:
    """bool masks [out, in] of the len(hidden_features)+1 masked linear layers."""
    d = int(order.numel())
    masks = []
    prev = None
    for i, width in enumerate(hidden_features):
        if width < d - 1:
            raise NotImplementedError("hidden width smaller than features-1 is not supported")
        cls = hidden_classes(d, width)
        if i == 0:
            masks.append(order[None, :] < cls[:, None])
        else:
            masks.append(prev[None, :] <= cls[:, None])
        prev = cls
    out_order = torch.repeat_interleave(order, total)
    masks.append(prev[None, :] <= out_order[:, None])
    return masks
