"""Discrepancy functions — mirrors mentflow/loss.py:7-17 (SlicedWassersteinDistance, an eval-only metric on a
third-party OT solver, is out of scope: SURVEY.md §2 row 8)."""
import torch

from . import ops


def _discrepancy(pred: torch.Tensor, targ: torch.Tensor, kind: str, pad: float, batch_div: float) -> torch.Tensor:
    S = pred.reshape(1, -1)
    _, D = ops.HistNormDiscFn.apply(S, targ.reshape(1, -1), False, 1.0, 1.0, 0.0, ops.DISCREPANCY_KINDS[kind], pad,
                                    float(batch_div))
    return D[0]


def mean_absolute_error(pred: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
    return _discrepancy(pred, targ, "mae", 0.0, pred.numel())


def mean_square_error(pred: torch.Tensor, targ: torch.Tensor) -> torch.Tensor:
    return _discrepancy(pred, targ, "mse", 0.0, pred.numel())


def kl_divergence(pred: torch.Tensor, targ: torch.Tensor, pad=1.00e-12) -> torch.Tensor:
    """F.kl_div(log(pred + pad), targ, reduction="batchmean"): sum / pred.shape[0]  (loss.py:15-17)."""
    return _discrepancy(pred, targ, "kld", pad, pred.shape[0])


kl_divergence.kind = "kld"
mean_absolute_error.kind = "mae"
mean_square_error.kind = "mse"
