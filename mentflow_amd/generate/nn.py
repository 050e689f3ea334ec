"""Density-free neural-network generator: the paper's "NN" baseline (interface of mentflow/generate/nn.py:18-85,
configured by experiments/config/gen/nn.yaml + model/nn.yaml: `entropy_estimator: none`, MAE discrepancy).

Outside the kernel scope: the network is a stack of library GEMMs; what matters here is that its
samples enter the same fused kick / projection / KDE / discrepancy kernels as the flow's, and that
`MENTFlow.loss` copes with a generator whose `log_prob` is None (`EmptyEntropyEstimator`)."""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .base import GenerativeModel

_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh}


def get_activation(name: str) -> nn.Module:
    try:
        return _ACTIVATIONS[name]()
    except KeyError:
        raise ValueError(f"Invalid activation '{name}'") from None


class NNTransform(nn.Module):
    """MLP  base -> phase space:  Linear, act, [Linear, Dropout, act] x hidden_layers, Linear  (one shared activation
    module, as the reference builds it; `layers.<k>` parameter names follow from the Sequential positions)."""

    def __init__(self, input_features: int = 2, output_features: int = 2, hidden_layers: int = 2, hidden_units: int = 20,
                 dropout: float = 0.0, activation: str = "tanh") -> None:
        super().__init__()
        act = get_activation(activation)
        widths = [input_features] + [hidden_units] * (hidden_layers + 1)
        stack: List[nn.Module] = []
        for depth, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            stack.append(nn.Linear(fan_in, fan_out))
            if depth > 0:
                stack.append(nn.Dropout(dropout))
            stack.append(act)
        stack.append(nn.Linear(hidden_units, output_features))
        self.layers = nn.Sequential(*stack)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x)


class NNGenerator(GenerativeModel):
    """x = transform(z), z ~ N(0, I_base_features); no density."""

    def __init__(self, base_features: int, transform: nn.Module) -> None:
        super().__init__()
        self.base_features = int(base_features)
        self.transform = transform
        self.inject_z: Optional[torch.Tensor] = None      # parity tests: fixed base draw

    def _device(self) -> torch.device:
        return next(self.transform.parameters()).device

    def sample_base(self, n: int) -> torch.Tensor:
        if self.inject_z is not None:
            return self.inject_z
        return torch.randn(int(n), self.base_features, device=self._device())

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self.transform(z)

    def forward_steps(self, z: torch.Tensor) -> List[torch.Tensor]:
        return [z, self.forward(z)]

    def sample(self, n: int) -> torch.Tensor:
        return self.forward(self.sample_base(n))

    def sample_and_log_prob(self, n: int) -> Tuple[torch.Tensor, None]:
        return self.sample(n), None

    def log_prob(self, x: torch.Tensor) -> None:
        return None
