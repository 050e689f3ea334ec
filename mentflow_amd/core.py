"""MENT-Flow model — mirrors mentflow/core.py:18-161 (same constructor kwargs, attributes and return values)."""
from __future__ import annotations

import logging
from typing import Callable, Iterator, List, Optional, Tuple

import torch
import torch.nn as nn

from . import dist as mfdist
from . import ops
from .diagnostics import Histogram, Histogram1D, Histogram2D
from .entropy import EmptyEntropyEstimator, MonteCarloEntropyEstimator
from .generate import GenerativeModel
from .loss import kl_divergence
from .simulate import forward, group_measurements
from .simulate.simulate import apply_pre
from .utils import unravel

log = logging.getLogger("mentflow_amd.core")


class MENTFlow(nn.Module):
    """Generative maximum-entropy tomography solver (core.py:18-61)."""

    def __init__(self, transforms: List[nn.Module], diagnostics: List[List[nn.Module]],
                 measurements: List[List[torch.Tensor]], generator: GenerativeModel, prior, entropy_estimator: Callable,
                 discrepancy_function: Callable = kl_divergence, penalty_parameter: float = 10.0) -> None:
        super().__init__()
        self.transforms = transforms
        self.diagnostics = self.set_diagnostics(diagnostics)
        self.measurements = self.set_measurements(measurements)
        self.generator = generator
        self.entropy_estimator = entropy_estimator
        self.discrepancy_function = discrepancy_function
        self.penalty_parameter = penalty_parameter
        self._plan = None
        self._plan_key = None
        self._no_plan_key = None
        self._generic_logged = False

    def set_diagnostics(self, diagnostics):
        self.diagnostics = diagnostics
        if self.diagnostics is None:
            self.diagnostics = [[]]
        return self.diagnostics

    def set_measurements(self, measurements):
        self.measurements = measurements
        if self.measurements is None:
            self.measurements = [[]]
        return self.measurements

    # ------------------------------------------------------------------ reference API (core.py:75-93)
    def sample(self, size: int) -> torch.Tensor:
        return self.generator.sample(int(size))

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        return self.generator.log_prob(x)

    def sample_and_log_prob(self, size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.generator.sample_and_log_prob(size)

    def sample_and_entropy(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x, log_prob = self.sample_and_log_prob(n)
        H = self.entropy_estimator(x, log_prob)
        return (x, H)

    def discrepancy_vector(self, predictions: List[List[torch.Tensor]]) -> List[torch.Tensor]:
        return [self.discrepancy_function(pred, meas)
                for pred, meas in zip(unravel(predictions), unravel(self.measurements))]

    # ------------------------------------------------------------------ data-parallel gradient reduction
    def _wire_gradient_reduction(self) -> None:
        """The forward all-reduce of the histogram sums has the IDENTITY as its adjoint (mentflow_amd/dist.py), which
        is only right if the per-rank parameter gradients are summed afterwards.  Flow generators expose a
        ``grad_reduce`` slot called once on their flat gradient (one all-reduce per step); any other generator (the
        NN baseline) gets a post-accumulate hook per parameter that all-reduces ``p.grad``.  Wired once per generator."""
        gen = self.generator
        if getattr(gen, "_mf_dp_wired", False):
            return
        if "grad_reduce" in vars(gen) or hasattr(type(gen), "grad_reduce"):
            if gen.grad_reduce is None:
                gen.grad_reduce = mfdist.reduce_gradients_
        else:
            params = [p for p in gen.parameters() if p.requires_grad]
            if not params:
                raise NotImplementedError("data-parallel training needs a generator with trainable parameters")
            for p in params:
                p.register_post_accumulate_grad_hook(lambda q: mfdist.reduce_gradients_(q.grad))
        gen._mf_dp_wired = True

    # ------------------------------------------------------------------ fused step
    def _fused_plan(self):
        """Static launch plan of the fused loss: one entry per distinct diagnostic object (the reference shares ONE
        Histogram object across all transforms, experiments/setup.py:43-44)."""
        kind = getattr(self.discrepancy_function, "kind", None)
        if kind is None or not isinstance(self.entropy_estimator, (MonteCarloEntropyEstimator, EmptyEntropyEstimator)):
            return None
        key = (tuple(id(t) for t in self.transforms), tuple(id(d) for d in unravel(self.diagnostics)),
               tuple(id(m) for m in unravel(self.measurements)),
               tuple((getattr(d, "kde", None), bool(getattr(d, "noise", False)) and getattr(d, "noise_scale", 0.0) > 0.0)
                     for d in unravel(self.diagnostics)))
        if self._plan is not None and self._plan_key == key:
            return self._plan
        if self._no_plan_key == key:     # "no fused plan" is cached too: the generic loop re-derives nothing per call
            return None
        self._no_plan_key = key          # (cleared again below if a plan is found)
        groups, generic = group_measurements(self.transforms, self.diagnostics)
        if generic:                      # a transport / diagnostic outside the fused kernels: the generic loop of loss()
            return None
        order = {}
        pos = 0
        for i in range(len(self.transforms)):
            for j in range(len(self.diagnostics[i])):
                order[(i, j)] = pos
                pos += 1
        plan = []
        for diagnostic, pre, slots, rows in groups.values():
            if not diagnostic.kde or (diagnostic.noise and diagnostic.noise_scale > 0.0):
                return None
            stacked = [torch.stack([r[k] for r in rows]).to(torch.float32).contiguous() for k in range(len(rows[0]))]
            meas = torch.stack([self.measurements[i][j] for (i, j) in slots]).to(torch.float32)
            meas = meas.reshape(len(slots), -1).contiguous()
            plan.append((diagnostic, stacked, meas, [order[s] for s in slots], pre))
        self._plan, self._plan_key, self._no_plan_key = (plan, pos, kind), key, None
        return self._plan

    def loss(self, batch_size: int) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
        """L = H + mu * mean_p D_p from a fresh batch (core.py:95-117).  ``batch_size`` is the GLOBAL number of
        particles; under torch.distributed every rank samples its share and two all-reduces make (L, H, D) and the
        parameter gradients identical to the single-GPU values (mentflow_amd/dist.py)."""
        plan = self._fused_plan()
        if plan is None:
            # generic loop of the reference (core.py:113-117): any nn.Module transport, any diagnostic, any discrepancy
            # callable; simulate.forward still takes the fused kernels for the pairs they cover
            if not self._generic_logged:
                self._generic_logged = True
                log.info("MENTFlow.loss: generic path (sample -> simulate.forward -> discrepancy_function per measurement)")
            if mfdist.is_active():
                return self._loss_generic_data_parallel(int(batch_size))
            x, H = self.sample_and_entropy(batch_size)
            predictions = forward(x, self.transforms, self.diagnostics)
            D = self.discrepancy_vector(predictions)
            L = H + self.penalty_parameter * (sum(D) / len(D))
            return (L, H, D)

        groups, n_meas, kind = plan
        n_total = int(batch_size)
        n_local = mfdist.local_batch(n_total)
        if mfdist.is_active():
            self._wire_gradient_reduction()
        x, log_prob = self.generator.sample_and_log_prob(n_local)

        use_entropy = isinstance(self.entropy_estimator, MonteCarloEntropyEstimator)
        if use_entropy and log_prob is None:
            raise ValueError("the Monte-Carlo entropy estimator needs a generator with a density (log_prob is None)")
        pieces = []
        for diagnostic, rows, meas, _, pre in groups:
            xt = apply_pre(x, pre)
            if isinstance(diagnostic, Histogram1D):
                S = ops.ProjKde1dFn.apply(xt, rows[0], diagnostic.coords, diagnostic.bandwidth_value,
                                          ops.kde_radius(diagnostic.bandwidth_bins))
            else:
                S = ops.ProjKde2dFn.apply(xt, rows[0], rows[1], diagnostic.coords_x, diagnostic.coords_y,
                                          diagnostic.bandwidth_values[0], diagnostic.bandwidth_values[1],
                                          ops.kde_radius(diagnostic.bandwidth_bins[0]),
                                          ops.kde_radius(diagnostic.bandwidth_bins[1]))
            pieces.append(S.reshape(-1))
        if use_entropy:
            pieces.append(ops.EntropySumsFn.apply(x, log_prob))
        if mfdist.is_active():
            flat = mfdist.all_reduce_sum(torch.cat(pieces))
            sizes = [p.numel() for p in pieces]
            pieces = list(torch.split(flat, sizes))

        D_slots: List[Optional[torch.Tensor]] = [None] * n_meas
        D_sum = None
        for (diagnostic, rows, meas, positions, _pre), S in zip(groups, pieces):
            P = meas.shape[0]
            if isinstance(diagnostic, Histogram1D):
                pre_scale, cell, div = 1.0 / n_total, diagnostic.resolution_value, float(meas.shape[1])
            else:
                pre_scale, cell = 1.0, diagnostic.resolution_values[0] * diagnostic.resolution_values[1]
                div = float(diagnostic.coords_x.numel())
            if kind != "kld":
                div = float(meas.shape[1])
            _, Dg = ops.HistNormDiscFn.apply(S.reshape(P, -1), meas, True, pre_scale, cell, 1.0e-10,
                                             ops.DISCREPANCY_KINDS[kind], 1.0e-12 if kind == "kld" else 0.0, div)
            for pos, dval in zip(positions, Dg.unbind(0)):
                D_slots[pos] = dval
            D_sum = Dg.sum() if D_sum is None else D_sum + Dg.sum()
        if use_entropy:
            H = self.entropy_estimator.from_sums(pieces[-1], n_total)
        else:
            H = torch.zeros((), dtype=torch.float32, device=x.device)
        L = H + self.penalty_parameter * (D_sum / n_meas)
        return (L, H, D_slots)

    def _loss_generic_data_parallel(self, n_total: int):
        """The generic loop (core.py:113-117) with the particle batch sharded over the ranks: every histogram diagnostic is a
        sum over particles followed by a normalisation, so the ranks all-reduce the raw sums of ALL measurements (and the two
        entropy sums) in ONE buffer and evaluate normalisation, discrepancy callable and loss redundantly — (L, H, D) are
        the single-process values on the concatenated batch, identical on every rank.  A diagnostic or an entropy estimator
        without a sum form raises, naming itself."""
        from .simulate.simulate import raw_sums
        self._wire_gradient_reduction()
        x, log_prob = self.generator.sample_and_log_prob(mfdist.local_batch(n_total))
        est = self.entropy_estimator
        use_entropy = isinstance(est, MonteCarloEntropyEstimator)
        if not use_entropy and not isinstance(est, EmptyEntropyEstimator):
            raise NotImplementedError(f"data-parallel loss: {type(est).__name__} has no sum form to reduce over the ranks "
                                      "(MonteCarloEntropyEstimator and EmptyEntropyEstimator do)")
        if use_entropy and log_prob is None:
            raise ValueError("the Monte-Carlo entropy estimator needs a generator with a density (log_prob is None)")
        pieces, finish = raw_sums(x, self.transforms, self.diagnostics)
        shapes = [p.shape for p in pieces]
        flat = [p.reshape(-1) for p in pieces]
        if use_entropy:
            flat.append(ops.EntropySumsFn.apply(x, log_prob))
        reduced = torch.split(mfdist.all_reduce_sum(torch.cat(flat)), [f.numel() for f in flat])
        predictions = finish([r.reshape(sh) for r, sh in zip(reduced, shapes)], n_total)
        D = self.discrepancy_vector(predictions)
        H = est.from_sums(reduced[-1], n_total) if use_entropy else torch.zeros((), dtype=torch.float32, device=x.device)
        L = H + self.penalty_parameter * (sum(D) / len(D))
        return (L, H, D)

    # ------------------------------------------------------------------ parameters / checkpoints (core.py:119-159)
    def parameters(self) -> Iterator[nn.Parameter]:
        return self.generator.parameters()

    def save(self, path) -> None:
        state = {
            "generator": self.generator.state_dict(),
            "entropy_estimator": self.entropy_estimator,
            "transforms": self.transforms,
            "diagnostics": self.diagnostics,
            "measurements": self.measurements,
        }
        torch.save(state, path)

    def load(self, path, device=None):
        state = torch.load(path, map_location=device, weights_only=False)
        try:
            self.generator.load_state_dict(state["generator"])
        except RuntimeError:
            raise RuntimeError("Error loading generative model. Architecture mismatch?")
        self.entropy_estimator = state["entropy_estimator"]
        self.transforms = state["transforms"]
        self.diagnostics = state["diagnostics"]
        self.measurements = state["measurements"]
        self.to(device)

    def to(self, device):
        if self.transforms is not None:
            for transform in self.transforms:
                transform.to(device)
        if self.diagnostics is not None:
            for index in range(len(self.diagnostics)):
                for diagnostic in self.diagnostics[index]:
                    diagnostic.to(device)
        if self.measurements is not None:
            for index in range(len(self.measurements)):
                self.measurements[index] = [m.to(device) for m in self.measurements[index]]
        if self.generator is not None:
            self.generator = self.generator.to(device)
        prior = getattr(self.entropy_estimator, "prior", None)
        if prior is not None and hasattr(prior, "to"):
            prior.to(device)
        self._plan = self._plan_key = self._no_plan_key = None
        return self
