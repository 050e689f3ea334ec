"""Plain neural-network generator — mirrors mentflow/generate/nn.py:18-85 (the paper's "NN" baseline model: no density,
used with `entropy_estimator: none` and the MAE discrepancy, experiments/config/model/nn.yaml).  It is an ordinary
`nn.Sequential` of library GEMMs (outside the kernel scope); its samples feed the same fused
projection + KDE + discrepancy kernels as the flows."""
from typing import Any, Callable, List, Tuple

import torch
import torch.nn as nn

from .base import GenerativeModel


def get_activation(name: str) -> Callable:
    if name == "relu":
        return nn.ReLU()
    elif name == "tanh":
        return nn.Tanh()
    raise ValueError(f"Invalid activation '{name}'")


class NNTransform(nn.Module):
    def __init__(self, input_features: int = 2, output_features: int = 2, hidden_layers: int = 2, hidden_units: int = 20,
                 dropout: float = 0.0, activation: str = "tanh") -> None:
        activation = get_activation(activation)
        super().__init__()
        layers = [nn.Linear(input_features, hidden_units), activation]
        for _ in range(hidden_layers):
            layers.append(nn.Linear(hidden_units, hidden_units))
            layers.append(nn.Dropout(dropout))
            layers.append(activation)
        layers.append(nn.Linear(hidden_units, output_features))
        self.layers = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x)


class NNGenerator(GenerativeModel):
    def __init__(self, base_features: int, transform: nn.Module) -> None:
        super().__init__()
        self.base_features = int(base_features)
        self.transform = transform
        self.inject_z = None                 # tests: fixed base draw

    def sample_base(self, n: int) -> torch.Tensor:
        if self.inject_z is not None:
            return self.inject_z
        dev = next(self.transform.parameters()).device
        return torch.randn((int(n), self.base_features), device=dev)          # MultivariateNormal(0, I).rsample

    def sample(self, n: int) -> torch.Tensor:
        return self.transform(self.sample_base(n))

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return None

    def sample_and_log_prob(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return (self.sample(n), None)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self.transform(z)

    def forward_steps(self, z: torch.Tensor) -> List[torch.Tensor]:
        return [z, self.transform(z)]
