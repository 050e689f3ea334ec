"""Generator factory — mirrors mentflow/generate/build.py:13-46,80-123."""
from __future__ import annotations

import torch

from .base import GenerativeModel
from .flows import AutoregressiveFlow
from .nn import NNGenerator, NNTransform

_REFERENCE_FLOW_NAMES = ["bpf", "ffjord", "gf", "gmm", "maf", "nag", "nsf", "sospf", "unaf"]


def build_flow(name: str, input_features: int, output_features: int, hidden_layers: int, hidden_units: int,
               transforms: int, device=None, **kws) -> AutoregressiveFlow:
    """build.py:13-46: features=output_features, hidden_features=hidden_layers*[hidden_units], `transforms` layers;
    "nsf" -> rational-quadratic spline (zuko default bins=8; experiments/setup.py:120-121 passes bins=20),
    "maf" -> affine.  Both are inverted (build.py:42-43): sampling is the single-pass direction."""
    if name == "nsf":
        flow = AutoregressiveFlow(output_features, hidden_layers * [hidden_units], transforms, "rqs", kws.pop("bins", 8))
    elif name == "maf":
        flow = AutoregressiveFlow(output_features, hidden_layers * [hidden_units], transforms, "affine")
    else:
        raise NotImplementedError(
            f"flow '{name}' is outside the MI355X hot-path scope (built: 'nsf', 'maf'; SURVEY.md §2 row 2)")
    if kws:
        raise TypeError(f"unexpected keyword arguments for '{name}': {sorted(kws)}")
    return flow.to(device)


def build_nn(input_features: int, output_features: int, hidden_layers: int, hidden_units: int, dropout: float = 0.0,
             activation: str = "tanh", device=None) -> NNGenerator:
    """build.py:49-77."""
    transform = NNTransform(input_features=input_features, output_features=output_features, hidden_layers=hidden_layers,
                            hidden_units=hidden_units, dropout=dropout, activation=activation)
    return NNGenerator(input_features, transform).to(device)


def build_generator(name: str, device: torch.device = None, **kws) -> GenerativeModel:
    """build.py:80-123."""
    if name == "nn":
        return build_nn(device=device, **kws)
    if name in _REFERENCE_FLOW_NAMES:
        return build_flow(name=name, device=device, **kws)
    raise ValueError(f"Invalid generative model name '{name}'")
