"""ORACLE (test infrastructure) — CPU restatement of the normalizing flow used by MENT-Flow.

*** parity unpinned *** : the arithmetic restated here lives in ``zuko==1.3.1`` (third party,
pinned by the reference's ``pyproject.toml:11``; source absent from /root/reference).  The
reference reaches it through ``mentflow/generate/build.py:13-46`` (``build_flow``: constructor
table, ``features/hidden_features/transforms`` kwargs, inversion of maf/nsf at :42-43) and
``mentflow/generate/flows/zuko.py:10-53`` (``WrappedZukoFlow``).  What follows restates zuko's
published algorithm (``zuko.flows.autoregressive.MAF / MaskedAutoregressiveTransform``,
``zuko.flows.spline.NSF``, ``zuko.nn.MaskedMLP / MaskedLinear``,
``zuko.transforms.MonotonicRQSTransform / MonotonicAffineTransform / AutoregressiveTransform /
ComposedTransform``, ``zuko.distributions.NormalizingFlow / DiagNormal``), see SURVEY.md App. A.

Everything is dtype-generic (run it in fp64 for gradcheck, fp32 for parity with the HIP path).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BOUND = 5.0
SLOPE = 1e-3


# ----------------------------------------------------------------------------------------------
# masks  (zuko.flows.autoregressive.MaskedAutoregressiveTransform.__init__, zuko.nn.MaskedMLP.__init__)
# ----------------------------------------------------------------------------------------------
def ar_adjacency(order: torch.Tensor, total: int) -> torch.Tensor:
    """adjacency[i*total + j, k] = order[i] > order[k]   (strict; passes == features).

    zuko: ``in_order = order``; ``out_order = repeat_interleave(order, total)``;
    ``adjacency = out_order[:, None] > in_order``.
    """
    out_order = torch.repeat_interleave(order, total)
    return out_order[:, None] > order[None, :]


def masked_mlp_masks(adjacency: torch.Tensor, hidden_features: Sequence[int]) -> List[torch.Tensor]:
    """Masks of every MaskedLinear of a zuko MaskedMLP (list of bool [out, in]).

    Follows zuko.nn.MaskedMLP.__init__ literally: unique rows, precedence matrix
    ``P_ij = (A A^T)_ij == |A_j|``, hidden units cycle over the reachable rows.
    """
    out_features, _ = adjacency.shape
    adj_u, inverse = torch.unique(adjacency, dim=0, return_inverse=True)
    precedence = adj_u.int() @ adj_u.int().t() == adj_u.sum(dim=-1)
    masks = []
    indices = None
    for i, features in enumerate((*hidden_features, out_features)):
        mask = precedence[:, indices] if i > 0 else adj_u
        if (~mask).all():
            raise ValueError("The adjacency matrix leads to a null Jacobian.")
        if i < len(hidden_features):
            reachable = mask.sum(dim=-1).nonzero().squeeze(dim=-1)
            indices = reachable[torch.arange(features) % len(reachable)]
            mask = mask[indices]
        else:
            mask = mask[inverse]
        masks.append(mask)
    return masks


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
@dataclass
class ARLayer:
    order: torch.Tensor                 # [d] long
    weights: List[torch.Tensor]         # len = hidden_layers + 1, each [out, in]
    biases: List[torch.Tensor]
    masks: List[torch.Tensor]           # bool, same shapes as weights


@dataclass
class FlowSpec:
    """A zuko MAF/NSF with `transforms` autoregressive layers, *inverted* as mentflow does
    (build.py:42-43), so that sampling is the single-pass direction."""
    features: int
    kind: str                           # "rqs" (nsf) | "affine" (maf)
    bins: int                           # K (rqs only)
    layers: List[ARLayer] = field(default_factory=list)

    @property
    def total(self) -> int:             # params per feature (zuko `total`)
        return 3 * self.bins - 1 if self.kind == "rqs" else 2

    def parameters(self) -> List[torch.Tensor]:
        out = []
        for layer in self.layers:
            for w, b in zip(layer.weights, layer.biases):
                out += [w, b]
        return out

    def to(self, dtype=None) -> "FlowSpec":
        new = FlowSpec(self.features, self.kind, self.bins)
        for l in self.layers:
            new.layers.append(
                ARLayer(
                    l.order.clone(),
                    [w.detach().clone().to(dtype) for w in l.weights],
                    [b.detach().clone().to(dtype) for b in l.biases],
                    [m.clone() for m in l.masks],
                )
            )
        return new


def init_flow(
    features: int,
    hidden_features: Sequence[int] = (64, 64, 64),
    transforms: int = 5,
    kind: str = "rqs",
    bins: int = 20,
    seed: Optional[int] = None,
    dtype=torch.float32,
) -> FlowSpec:
    """zuko.flows.MAF.__init__ / NSF.__init__: `transforms` layers, order = arange for even
    layers, reversed for odd layers; each hyper-network is MaskedMLP(adjacency, hidden_features)
    built from torch.nn.Linear (default init, construction order layer by layer)."""
    if seed is not None:
        torch.manual_seed(seed)
    spec = FlowSpec(features, kind, bins)
    orders = [torch.arange(features), torch.flipud(torch.arange(features))]
    for t in range(transforms):
        order = orders[t % 2].clone()
        adjacency = ar_adjacency(order, spec.total)
        masks = masked_mlp_masks(adjacency, hidden_features)
        ws, bs = [], []
        for m in masks:
            lin = torch.nn.Linear(m.shape[1], m.shape[0])
            ws.append(lin.weight.detach().clone().to(dtype))
            bs.append(lin.bias.detach().clone().to(dtype))
        spec.layers.append(ARLayer(order, ws, bs, masks))
    return spec


# ----------------------------------------------------------------------------------------------
# conditioner (zuko.nn.MaskedLinear.forward = F.linear(x, mask * W, b); ReLU between)
# ----------------------------------------------------------------------------------------------
def conditioner(x: torch.Tensor, layer: ARLayer, total: int) -> torch.Tensor:
    h = x
    n = len(layer.weights)
    for i, (w, b, m) in enumerate(zip(layer.weights, layer.biases, layer.masks)):
        h = F.linear(h, m.to(w.dtype) * w, b)
        if i < n - 1:
            h = torch.relu(h)
    return h.unflatten(-1, (-1, total))          # [N, d, total]


# ----------------------------------------------------------------------------------------------
# univariate transforms
# ----------------------------------------------------------------------------------------------
def rqs_knots(phi: torch.Tensor, bins: int, bound: float = BOUND, slope: float = SLOPE):
    """zuko.transforms.MonotonicRQSTransform.__init__: phi [..., 3K-1] -> knots X,Y [..., K+1], D [..., K+1]."""
    w, h, d = phi.split((bins, bins, bins - 1), dim=-1)
    ls = math.log(slope)
    w = w / (1 + abs(2 * w / ls))
    h = h / (1 + abs(2 * h / ls))
    d = d / (1 + abs(d / ls))
    w = F.pad(F.softmax(w, dim=-1), (1, 0), value=0)
    h = F.pad(F.softmax(h, dim=-1), (1, 0), value=0)
    d = F.pad(d, (1, 1), value=0)
    X = bound * (2 * torch.cumsum(w, dim=-1) - 1)
    Y = bound * (2 * torch.cumsum(h, dim=-1) - 1)
    D = torch.exp(d)
    return X, Y, D


def _searchsorted(seq: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    return torch.sum(seq < value[..., None], dim=-1)


def _rqs_bin(X, Y, D, k, bins):
    mask = torch.logical_and(0 <= k, k < bins)
    k = k % bins
    k0 = k[..., None]
    k1 = k0 + 1
    x0, x1 = X.gather(-1, k0).squeeze(-1), X.gather(-1, k1).squeeze(-1)
    y0, y1 = Y.gather(-1, k0).squeeze(-1), Y.gather(-1, k1).squeeze(-1)
    d0, d1 = D.gather(-1, k0).squeeze(-1), D.gather(-1, k1).squeeze(-1)
    s = (y1 - y0) / (x1 - x0)
    return mask, x0, x1, y0, y1, d0, d1, s


def rqs_forward(x: torch.Tensor, phi: torch.Tensor, bins: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """MonotonicRQSTransform.call_and_ladj: returns (y, log|dy/dx|) elementwise."""
    X, Y, D = rqs_knots(phi, bins)
    k = _searchsorted(X, x) - 1
    mask, x0, x1, y0, y1, d0, d1, s = _rqs_bin(X, Y, D, k, bins)
    z = mask * (x - x0) / (x1 - x0)
    y = y0 + (y1 - y0) * (s * z**2 + d0 * z * (1 - z)) / (s + (d0 + d1 - 2 * s) * z * (1 - z))
    jac = (
        s**2
        * (2 * s * z * (1 - z) + d0 * (1 - z) ** 2 + d1 * z**2)
        / (s + (d0 + d1 - 2 * s) * z * (1 - z)) ** 2
    )
    return torch.where(mask, y, x), mask * jac.log()


def rqs_inverse(y: torch.Tensor, phi: torch.Tensor, bins: int) -> torch.Tensor:
    """MonotonicRQSTransform._inverse."""
    X, Y, D = rqs_knots(phi, bins)
    k = _searchsorted(Y, y) - 1
    mask, x0, x1, y0, y1, d0, d1, s = _rqs_bin(X, Y, D, k, bins)
    y_ = mask * (y - y0)
    a = (y1 - y0) * (s - d0) + y_ * (d0 + d1 - 2 * s)
    b = (y1 - y0) * d0 - y_ * (d0 + d1 - 2 * s)
    c = -s * y_
    z = 2 * c / (-b - (b**2 - 4 * a * c).sqrt())
    x = x0 + z * (x1 - x0)
    return torch.where(mask, x, y)


def affine_log_scale(scale: torch.Tensor, slope: float = SLOPE) -> torch.Tensor:
    """MonotonicAffineTransform.__init__: log_scale = scale / (1 + |scale / log(slope)|)."""
    return scale / (1 + abs(scale / math.log(slope)))


def affine_forward(x, phi):
    shift, scale = phi[..., 0], phi[..., 1]
    ls = affine_log_scale(scale)
    return x * ls.exp() + shift, ls


def affine_inverse(y, phi):
    shift, scale = phi[..., 0], phi[..., 1]
    ls = affine_log_scale(scale)
    return (y - shift) * (-ls).exp()


# ----------------------------------------------------------------------------------------------
# autoregressive layer, composed flow
# ----------------------------------------------------------------------------------------------
def ar_call_and_ladj(x: torch.Tensor, layer: ARLayer, spec: FlowSpec) -> Tuple[torch.Tensor, torch.Tensor]:
    """AutoregressiveTransform.call_and_ladj: y = meta(x)(x) (one conditioner pass), ladj summed over features
    (DependentTransform(..., 1))."""
    phi = conditioner(x, layer, spec.total)
    if spec.kind == "rqs":
        y, ladj = rqs_forward(x, phi, spec.bins)
    else:
        y, ladj = affine_forward(x, phi)
    return y, ladj.sum(dim=-1)


def ar_inverse(y: torch.Tensor, layer: ARLayer, spec: FlowSpec) -> torch.Tensor:
    """AutoregressiveTransform._inverse: x = 0; repeat `passes`(=d) times: x = meta(x).inv(y)."""
    x = torch.zeros_like(y)
    for _ in range(spec.features):
        phi = conditioner(x, layer, spec.total)
        x = rqs_inverse(y, phi, spec.bins) if spec.kind == "rqs" else affine_inverse(y, phi)
    return x


def flow_forward(z: torch.Tensor, spec: FlowSpec) -> Tuple[torch.Tensor, torch.Tensor]:
    """ComposedTransform.call_and_ladj over the layers in list order (the *sampling* direction of the
    inverted flow: WrappedZukoFlow.forward, mentflow/generate/flows/zuko.py:28-29)."""
    x = z
    ladj = torch.zeros(z.shape[:-1], dtype=z.dtype)
    for layer in spec.layers:
        x, l = ar_call_and_ladj(x, layer, spec)
        ladj = ladj + l
    return x, ladj


def flow_forward_steps(z: torch.Tensor, spec: FlowSpec) -> List[torch.Tensor]:
    """WrappedZukoFlow.forward_steps (flows/zuko.py:34-41)."""
    xs = [z.clone()]
    x = z
    for layer in spec.layers:
        x, _ = ar_call_and_ladj(x, layer, spec)
        xs.append(x)
    return xs


def flow_inverse(x: torch.Tensor, spec: FlowSpec) -> torch.Tensor:
    """WrappedZukoFlow.inverse (flows/zuko.py:31-32): layers in reverse, d passes each."""
    z = x
    for layer in reversed(spec.layers):
        z = ar_inverse(z, layer, spec)
    return z


def flow_inverse_steps(x: torch.Tensor, spec: FlowSpec) -> List[torch.Tensor]:
    """WrappedZukoFlow.inverse_steps (flows/zuko.py:43-50)."""
    zs = [x.clone()]
    z = x
    for layer in reversed(spec.layers):
        z = ar_inverse(z, layer, spec)
        zs.append(z)
    return zs


def base_log_prob(z: torch.Tensor) -> torch.Tensor:
    """zuko DiagNormal(0, 1).log_prob: -1/2 sum z^2 - d/2 log(2 pi)."""
    d = z.shape[-1]
    return -0.5 * (z**2).sum(dim=-1) - 0.5 * d * math.log(2 * math.pi)


def sample_and_log_prob(z: torch.Tensor, spec: FlowSpec) -> Tuple[torch.Tensor, torch.Tensor]:
    """NormalizingFlow.rsample_and_log_prob with the base draw `z` injected
    (WrappedZukoFlow.sample_and_log_prob, flows/zuko.py:24-26): x = F(z), logp = logN(z) - ladj."""
    x, ladj = flow_forward(z, spec)
    return x, base_log_prob(z) - ladj


def log_prob(x: torch.Tensor, spec: FlowSpec) -> torch.Tensor:
    """NormalizingFlow.log_prob (WrappedZukoFlow.log_prob, flows/zuko.py:21-22):
    z = F^-1(x); logp = logN(z) - ladj_F(z)."""
    z = flow_inverse(x, spec)
    _, ladj = flow_forward(z, spec)
    return base_log_prob(z) - ladj
