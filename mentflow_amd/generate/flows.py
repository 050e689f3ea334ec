"""Autoregressive normalizing flows (NSF / MAF) on the gfx950 kernels.

Drop-in for ``mentflow.generate.flows.WrappedZukoFlow`` (mentflow/generate/flows/zuko.py:10-53) wrapping a zuko
``NSF``/``MAF`` inverted as in mentflow/generate/build.py:42-43.  The module tree reproduces zuko 1.3.1's, so that
``state_dict()`` keys have the same names ([MEM] — zuko is not available to check):

    _flow.transform.transform.transforms.{t}.hyper.{0,2,4,...}.{weight,bias,mask}
    _flow.transform.transform.transforms.{t}.order
    _flow.base._0, _flow.base._1

Parameters are ordinary ``nn.Linear`` weights (same default init, same construction order as zuko's MaskedMLP), but
their storage is ONE flat buffer (each ``weight`` / ``bias`` is a view into it, each ``.grad`` a view into one flat
gradient buffer): a call packs the flat buffer (one gather kernel) into the per-layer LDS images the kernels consume,
and the backward deposits the flat gradient with one copy instead of 40 per-parameter accumulation kernels — at the
reference's 25 000-particle batch that host / launch overhead is a third of the step.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import get_lib
from .base import GenerativeModel
from .masks import conditioner_masks
from . import packing


class MaskedLinear(nn.Linear):
    """zuko.nn.MaskedLinear: linear layer whose weight is multiplied by a constant 0/1 mask."""

    def __init__(self, adjacency: torch.Tensor):
        super().__init__(adjacency.shape[1], adjacency.shape[0])
        self.register_buffer("mask", adjacency)


class MaskedAutoregressiveTransform(nn.Module):
    """Parameter holder of one autoregressive layer (zuko MaskedAutoregressiveTransform)."""

    def __init__(self, features: int, order: torch.Tensor, hidden_features: Sequence[int], total: int):
        super().__init__()
        self.register_buffer("order", order)
        masks = conditioner_masks(order, hidden_features, total)
        layers: List[nn.Module] = []
        for i, m in enumerate(masks):
            layers.append(MaskedLinear(m))
            if i < len(masks) - 1:
                layers.append(nn.ReLU())
        self.hyper = nn.Sequential(*layers)

    def linears(self) -> List[MaskedLinear]:
        return [m for m in self.hyper if isinstance(m, MaskedLinear)]


class _Composed(nn.Module):
    def __init__(self, transforms: Sequence[nn.Module]):
        super().__init__()
        self.transforms = nn.ModuleList(transforms)


class _Inverse(nn.Module):
    def __init__(self, transform: nn.Module):
        super().__init__()
        self.transform = transform


class _DiagNormalBase(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.register_buffer("_0", torch.zeros(features))
        self.register_buffer("_1", torch.ones(features))


class _Flow(nn.Module):
    def __init__(self, transform: nn.Module, base: nn.Module):
        super().__init__()
        self.transform = transform
        self.base = base


class AutoregressiveFlow(GenerativeModel):
    """NSF ("rqs") or MAF ("affine") generator with the reference's ``GenerativeModel`` API.

    Autograd contract.  ``sample_and_log_prob`` / ``forward`` are differentiable in the base draw ``z`` (as zuko's
    transform is, flows/zuko.py:28-29) and in the parameters.  By default the PARAMETERS are not inputs of the autograd
    node: the kernels read the flat buffer the parameters are views of, and the backward deposits the flat gradient
    straight into the parameters' ``.grad`` (one copy instead of 40 accumulation kernels).  Consequences:
    ``loss.backward()`` + any optimizer work as usual; ``torch.autograd.grad(loss, generator.parameters())`` raises
    ("not used in the graph"); ``autograd.grad(loss, other_inputs)`` still fills the parameters' ``.grad`` as a side
    effect; per-parameter gradient hooks do not fire.  Set ``autograd_parameters = True`` to make the parameters ordinary
    autograd inputs (one ``torch.cat`` per call and one accumulation per parameter: slower at small batches, identical
    values) when those features are needed.  tests/test_flow_autograd_contract.py pins both behaviours."""

    def __init__(self, features: int, hidden_features: Sequence[int] = (64, 64, 64), transforms: int = 5,
                 kind: str = "rqs", bins: int = 20):
        super().__init__()
        if kind not in ("rqs", "affine"):
            raise ValueError(kind)
        widths = {int(h) for h in hidden_features}
        if len(widths) != 1 or not 1 <= max(widths) <= packing.WIDE_HP:
            raise NotImplementedError(
                f"the gfx950 flow kernels take one width for every hidden layer and hidden_units <= {packing.WIDE_HP} "
                f"(got {tuple(hidden_features)})")
        if features > packing.WIDE_DMAX:
            raise NotImplementedError(f"the gfx950 flow kernels take up to {packing.WIDE_DMAX} features (got {features})")
        if max(widths) < features - 1:
            raise NotImplementedError("hidden width smaller than features - 1 is not supported")
        # Two kernel families behind one module.  Up to 64 hidden units and 7 features: the tuned kernels that keep the layer's
        # weights in LDS (narrower layers ride zero-padded in the 64-wide image).  Beyond (hidden_units 65 .. 128 or 8 .. 16
        # features; mentflow/generate/build.py:36-38 takes both from the config): the wide family, weights in global memory as MFMA
        # fragment blocks (mentflow_amd/csrc/flow_wide.hip) — the 128-wide last layer alone is 198 KB at d = 6, against 160 KB of LDS.
        self.wide = max(widths) > packing.HID or features > 7 or os.environ.get("MENTFLOW_FORCE_WIDE", "") == "1"   # (env: A/B runs)
        self.features, self.kind, self.bins = int(features), kind, int(bins)
        self.hidden_features = tuple(int(h) for h in hidden_features)
        self.total = 3 * self.bins - 1 if kind == "rqs" else 2
        orders = [torch.arange(features), torch.flipud(torch.arange(features))]
        layers = [MaskedAutoregressiveTransform(features, orders[t % 2].clone(), self.hidden_features, self.total)
                  for t in range(transforms)]
        self._flow = _Flow(_Inverse(_Composed(layers)), _DiagNormalBase(features))
        self._spec: Optional[ops.FlowSpec] = None
        self._spec_device = None
        self.grad_reduce = None          # set by mentflow_amd.dist for data-parallel runs
        self.autograd_parameters = False # True: parameters are autograd inputs (see the class docstring)
        self._flat = None                # flat parameter / gradient buffers (see _flatten)
        self._gflat = None
        self._gviews = None
        self._flat_params = None
        # 0-dim leaf that makes autograd call the flow's backward; not registered: invisible to parameters() / state_dict()
        self._trigger = [torch.zeros((), requires_grad=True)]
        self.inject_z: Optional[torch.Tensor] = None   # parity tests: base draw used instead of a fresh one

    # ------------------------------------------------------------------ packing
    @property
    def layers(self) -> List[MaskedAutoregressiveTransform]:
        return list(self._flow.transform.transform.transforms)

    # ------------------------------------------------------------------ flat parameter / gradient storage
    def _flat_ok(self) -> bool:
        """True iff EVERY current parameter is still the view of the flat buffer that _flatten made: same Parameter
        objects in the same order, same storage offsets, float32, same device.  (~40 host compares per call.  Checking
        only the first and last parameter missed ``ps[3].data = ...`` or a swapped Parameter in the middle: the kernels
        then kept consuming the stale flat slice while parameters() / state_dict() / the optimizer saw the new tensor.)"""
        ps = self._flat_params
        f = self._flat
        if not ps or f is None:
            return False
        cur = list(self.parameters())
        if len(cur) != len(ps):
            return False
        base, off = f.data_ptr(), 0
        for p, q in zip(cur, ps):
            if p is not q or p.dtype != torch.float32 or p.device != f.device or p.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
        return off == f.numel()

    def _flatten(self) -> None:
        """(Re)build the flat buffers and point every parameter's storage at its slice.  Needed once, and again after
        anything that replaces the parameters' storage (``.to(device)``, ``.float()``)."""
        params = list(self.parameters())
        bad = [n for n, p in self.named_parameters() if p.dtype != torch.float32]
        if bad:
            raise TypeError(f"the gfx950 flow kernels compute in float32; parameter(s) {bad[:3]} are "
                            f"{next(p.dtype for p in params if p.dtype != torch.float32)} (e.g. after .double() / .half()): "
                            "convert the generator back with .float()")
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        views, off = [], 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                if p.grad is not None:
                    gflat[off:off + n].copy_(p.grad.reshape(-1))
                    p.grad = gflat[off:off + n].view(p.shape)
                views.append(gflat[off:off + n].view(p.shape))
                off += n
        self._flat, self._gflat, self._gviews, self._flat_params = flat, gflat, views, params

    def flat_parameters(self) -> torch.Tensor:
        """All parameters as one contiguous vector (the parameters ARE views of it; no copy)."""
        if not self._flat_ok():
            self._flatten()
        return self._flat

    def _deposit_gradient(self, gflat: torch.Tensor) -> None:
        """Called by the flow's backward with the flat parameter gradient: all-reduce across ranks (data-parallel runs),
        then ACCUMULATE into the parameters' ``.grad`` as autograd would — with one kernel when the gradients are unset
        (``zero_grad(set_to_none=True)``) or are this module's own views (``set_to_none=False``)."""
        if self.grad_reduce is not None:
            self.grad_reduce(gflat)
        params, views = self._flat_params, self._gviews
        if not any(p.requires_grad for p in params):
            return                                  # generator frozen as a whole (the call was for dL/dz only)
        if not all(p.requires_grad for p in params):
            raise NotImplementedError("freezing individual flow parameters is not supported (freeze the generator as a whole)")
        if all(p.grad is None for p in params):
            self._gflat.copy_(gflat)
            for p, v in zip(params, views):
                p.grad = v
        elif all(p.grad is v for p, v in zip(params, views)):
            self._gflat.add_(gflat)
        else:                                   # somebody installed their own .grad tensors: per-parameter accumulation
            off = 0
            for p in params:
                n = p.numel()
                g = gflat[off:off + n].view(p.shape)
                p.grad = g.clone() if p.grad is None else p.grad + g
                off += n

    def build_index_maps(self) -> Tuple[np.ndarray, np.ndarray, int]:
        """(image_index [T*image_floats], grad_index [numel], image_floats) — pure host logic."""
        d, L = self.features, len(self.hidden_features)
        offsets, off = [], 0
        for p in self.parameters():
            offsets.append(off)
            off += p.numel()
        numel = off
        per_layer = 2 * (L + 1)
        deriv_slot = None
        if self.kind == "rqs":
            deriv_slot = get_lib().mf_flow_rqs_deriv_slot(self.bins)
            if deriv_slot < 0:
                raise NotImplementedError(f"bins={self.bins}: the spline kernels take 2 <= bins <= 21 (32 slots per lane half)")
        idx = []
        if self.wide:
            nblk = d if self.kind == "rqs" else 1
            grad_floats = packing.wide_grad_layout(L, nblk)["total"]
            grad_index = np.full(numel, -1, dtype=np.int64)
            for t, layer in enumerate(self.layers):
                masks = [lin.mask.cpu() for lin in layer.linears()]
                img, gparam, gpos = packing.wide_image_index(d, L, self.kind, self.bins, masks,
                                                             offsets[t * per_layer:(t + 1) * per_layer], deriv_slot)
                idx.append(img)
                grad_index[gparam] = t * grad_floats + gpos
            self._grad_floats = grad_floats
            return np.concatenate(idx), grad_index.astype(np.int32), idx[0].size
        for t, layer in enumerate(self.layers):
            masks = [lin.mask.cpu() for lin in layer.linears()]
            idx.append(packing.layer_image_index(d, L, self.kind, self.bins, masks,
                                                 offsets[t * per_layer:(t + 1) * per_layer], deriv_slot))
        image_index = np.concatenate(idx)
        self._grad_floats = idx[0].size
        return image_index, packing.invert_index(image_index, numel), idx[0].size

    def spec(self) -> ops.FlowSpec:
        dev = next(self.parameters()).device
        if self._spec is None or self._spec_device != dev:
            image_index, grad_index, image_floats = self.build_index_maps()
            lib = get_lib()
            L = len(self.hidden_features)
            if self.wide:
                nblk = self.features if self.kind == "rqs" else 1
                expect, expect_g = lib.mf_flow_wide_image_floats(L, nblk), lib.mf_flow_wide_grad_floats(L, nblk)
            else:
                expect = (lib.mf_flow_image_floats(self.features, L) if self.kind == "rqs"
                          else lib.mf_flow_affine_image_floats(self.features, L))
                expect_g = expect
            if expect != image_floats or expect_g != self._grad_floats:
                raise RuntimeError(f"image layout mismatch: host {image_floats} / {self._grad_floats} vs library {expect} / {expect_g}")
            self._spec = ops.FlowSpec(self.features, L, len(self.layers), self.kind, self.bins,
                                      image_floats, torch.from_numpy(image_index).to(dev),
                                      torch.from_numpy(grad_index).to(dev),
                                      [layer.order.cpu().tolist() for layer in self.layers],
                                      wide=self.wide, hidden=self.hidden_features[0], grad_floats=self._grad_floats)
            self._spec_device = dev
        return self._spec

    # ------------------------------------------------------------------ GenerativeModel API (flows/zuko.py:15-53)
    def dim(self) -> int:
        return self.features

    def sample_base(self, n: int) -> torch.Tensor:
        dev = next(self.parameters()).device
        return torch.randn((int(n), self.features), device=dev)

    def sample_and_log_prob(self, n: int, z: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """x = F(z), log_prob = logN(z) - ladj.  ``z`` may be injected (parity tests); default: fresh base draw."""
        if z is None:
            z = self.inject_z if self.inject_z is not None else self.sample_base(n)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters())):
            flat = self.flat_parameters()
            if self.autograd_parameters:
                # parameters as autograd inputs: the gradient comes back through torch.cat (per-parameter accumulation)
                flat_ag = torch.cat([p.reshape(-1) for p in self.parameters()])
                return ops.FlowSampleFn.apply(z, flat_ag, self.spec(), self.grad_reduce, None, None)
            if self._trigger[0].device != flat.device:
                self._trigger[0] = torch.zeros((), device=flat.device, requires_grad=True)
            return ops.FlowSampleFn.apply(z, flat, self.spec(), None, self._trigger[0], self._deposit_gradient)
        xs, logp = ops.flow_layers_forward(z, self.flat_parameters(), self.spec())
        return xs[-1], logp

    def sample(self, n: int) -> torch.Tensor:
        return self.sample_and_log_prob(n)[0]

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self.sample_and_log_prob(z.shape[0], z=z)[0]

    def forward_steps(self, z: torch.Tensor) -> List[torch.Tensor]:
        with torch.no_grad():
            xs, _ = ops.flow_layers_forward(z.clone(), self.flat_parameters(), self.spec())
        return xs

    def inverse(self, x: torch.Tensor) -> torch.Tensor:
        """z = F^-1(x) (flows/zuko.py:31-32): layers in reverse, d autoregressive passes each.  No autograd."""
        with torch.no_grad():
            return ops.flow_layers_inverse(x, self.flat_parameters(), self.spec())[-1]

    def inverse_steps(self, x: torch.Tensor) -> List[torch.Tensor]:
        with torch.no_grad():
            return ops.flow_layers_inverse(x.clone(), self.flat_parameters(), self.spec())

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        """Density of arbitrary points (flows/zuko.py:21-22): z = F^-1(x), log_prob = logN(z) - ladj_F(z).
        Evaluation only (notebooks, eval): no autograd graph is recorded."""
        with torch.no_grad():
            z = ops.flow_layers_inverse(x, self.flat_parameters(), self.spec())[-1]
            return ops.flow_layers_forward(z, self.flat_parameters(), self.spec())[1]
