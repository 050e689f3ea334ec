"""torch.autograd wrappers over the C ABI of libmentflow_hip.so (include/mentflow_hip.h).

Each Function's forward AND backward is a hand-written gfx950 kernel; torch supplies device memory, the current
stream and the autograd graph between ops, nothing else.  All tensors must live on the GPU (mentflow_amd._lib.ptr
raises otherwise): there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

_F32 = torch.float32
_ENV_ACT_LEVEL = int(os.environ["MENTFLOW_ACT_LEVEL"]) if os.environ.get("MENTFLOW_ACT_LEVEL", "") != "" else None


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != _F32:
        raise RuntimeError(f"mentflow_amd kernels compute in float32 (got {t.dtype})")
    return t.contiguous()


def kde_radius(bandwidth_in_bins: float) -> int:
    """Truncation radius (bins) of the Gaussian KDE: weights beyond (R + 1/2) bins are < exp(-40.5) = 2.6e-18."""
    return max(1, int(math.ceil(9.0 * float(bandwidth_in_bins) - 0.5)))


# ------------------------------------------------------------------------------------------------ flow
class FlowSpec:
    """Static description of a packed flow (shared by every call): geometry + device index maps."""

    def __init__(self, d: int, hidden_layers: int, transforms: int, kind: str, bins: int, image_floats: int,
                 image_index: torch.Tensor, grad_index: torch.Tensor, orders, wide: bool = False, hidden: int = 64,
                 grad_floats: Optional[int] = None):
        self.d, self.L, self.T, self.kind, self.bins = d, hidden_layers, transforms, kind, bins
        # wide = the mf_flow_wide_* family (hidden_units 65 .. 128 and / or 8 .. 16 features: weights in global memory as MFMA
        # fragments, two-kernel backward, gradient slabs of `grad_floats` floats in natural order)
        self.wide, self.hidden = bool(wide), int(hidden)
        self.grad_floats = int(image_floats if grad_floats is None else grad_floats)
        # per layer: host int32 array of the autoregressive order (lets the kernels skip masked-out MFMA k-steps)
        self.orders = [(C.c_int32 * d)(*[int(v) for v in o]) for o in orders]
        self.sparse = True
        self.image_floats = image_floats
        self.image_index = image_index      # int32 [T * image_floats]  -> flat parameter index or -1
        self.grad_index = grad_index        # int32 [numel]             -> position in the image stack or -1
        self.bwd_chunk = 1 << 20            # particles per backward chunk (3 KiB of scratch each at d=6)
        # Activation hand-off from the training forward to the fused backward through HBM (mf_flow_rqs_layer_fwd_save):
        # None = the highest level (2: hidden tiles + conditioner outputs, 1: hidden tiles, 0: recompute everything) that the
        # kernels support for this flow AND whose buffers (T layers x 1 712 / 512 B per particle at d = 6, L = 3, 20 bins) fit
        # `act_budget_bytes`; an int pins the level (tests, A/B runs).  Environment: MENTFLOW_ACT_LEVEL.
        self.act_level: Optional[int] = _ENV_ACT_LEVEL
        self.act_budget_bytes: Optional[int] = None     # None: 60 % of the device's memory (173 GB on an MI355X)
        self._act_supported: Optional[int] = None

    def resolve_act_level(self, n: int, device: torch.device) -> int:
        if n <= 0:
            return 0
        lib = _lib.get_lib()
        if self.wide:
            # one level: hidden tiles + conditioner outputs (mf_flow_wide_layer_fwd_save), if T such buffers fit the budget
            want = 1 if self.act_level is None else min(int(self.act_level), 1)
            budget = self.act_budget_bytes
            if budget is None:
                budget = int(0.6 * torch.cuda.get_device_properties(device).total_memory) if device.type == "cuda" else 1 << 62
            bins = self.bins if self.kind == "rqs" else 0
            return want if want > 0 and 4 * self.T * lib.mf_flow_wide_act_floats(n, self.d, self.L, bins) <= budget else 0
        if self.kind != "rqs" or not self.sparse:
            return 0
        supported = min(lib.mf_flow_rqs_act_level(self.d, self.L, self.bins, o) for o in self.orders)
        want = supported if self.act_level is None else min(int(self.act_level), supported)
        budget = self.act_budget_bytes
        if budget is None:
            budget = int(0.6 * torch.cuda.get_device_properties(device).total_memory) if device.type == "cuda" else 1 << 62
        while want > 0 and 4 * self.T * lib.mf_flow_rqs_act_floats(n, self.d, self.L, self.bins, want) > budget:
            want -= 1
        return want


def _layer_fwd(spec: FlowSpec, t: int, image: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
               logp_in: Optional[torch.Tensor], logp_out: torch.Tensor, init: bool, act: Optional[torch.Tensor] = None,
               act_level: int = 0) -> None:
    n = x.shape[0]
    order = spec.orders[t] if spec.sparse else None
    if spec.wide and act_level > 0:
        call("mf_flow_wide_layer_fwd_save", ptr(image), spec.d, spec.hidden, spec.L, spec.bins if spec.kind == "rqs" else 0, order,
             ptr(x), n, ptr(y), ptr(logp_in), ptr(logp_out), int(init), ptr(act), act.numel(), stream_ptr(x))
    elif spec.wide:
        call("mf_flow_wide_layer_fwd", ptr(image), spec.d, spec.hidden, spec.L, spec.bins if spec.kind == "rqs" else 0, order,
             ptr(x), n, ptr(y), ptr(logp_in), ptr(logp_out), int(init), stream_ptr(x))
    elif act_level > 0:
        call("mf_flow_rqs_layer_fwd_save", ptr(image), spec.d, spec.L, spec.bins, order, ptr(x), n, ptr(y), ptr(logp_in),
             ptr(logp_out), int(init), ptr(act), act.numel(), int(act_level), stream_ptr(x))
    elif spec.kind == "rqs":
        call("mf_flow_rqs_layer_fwd", ptr(image), spec.d, spec.L, spec.bins, order, ptr(x), n, ptr(y), ptr(logp_in),
             ptr(logp_out), int(init), stream_ptr(x))
    else:
        call("mf_flow_affine_layer_fwd", ptr(image), spec.d, spec.L, order, ptr(x), n, ptr(y), ptr(logp_in),
             ptr(logp_out), int(init), stream_ptr(x))


def _layer_bwd(spec: FlowSpec, t: int, image, x, gy, glogp, gx, gslab, accumulate: bool, scratch, act=None,
               act_level: int = 0) -> None:
    n = x.shape[0]
    order = spec.orders[t] if spec.sparse else None
    rows = gslab.shape[0]
    if spec.wide and act_level > 0:
        call("mf_flow_wide_layer_bwd_saved", ptr(image), spec.d, spec.hidden, spec.L, spec.bins if spec.kind == "rqs" else 0, order,
             ptr(x), n, ptr(gy), ptr(glogp), ptr(gx), ptr(gslab), rows, int(accumulate), ptr(scratch), scratch.numel(), ptr(act),
             act.numel(), stream_ptr(x))
    elif spec.wide:
        call("mf_flow_wide_layer_bwd", ptr(image), spec.d, spec.hidden, spec.L, spec.bins if spec.kind == "rqs" else 0, order,
             ptr(x), n, ptr(gy), ptr(glogp), ptr(gx), ptr(gslab), rows, int(accumulate), ptr(scratch), scratch.numel(),
             stream_ptr(x))
    elif act_level > 0:
        call("mf_flow_rqs_layer_bwd_saved", ptr(image), spec.d, spec.L, spec.bins, order, ptr(x), n, ptr(gy), ptr(glogp),
             ptr(gx), ptr(gslab), rows, int(accumulate), ptr(act), act.numel(), int(act_level), stream_ptr(x))
    elif spec.kind == "rqs":
        call("mf_flow_rqs_layer_bwd", ptr(image), spec.d, spec.L, spec.bins, order, ptr(x), n, ptr(gy), ptr(glogp),
             ptr(gx), ptr(gslab), rows, int(accumulate), ptr(scratch), scratch.numel(), stream_ptr(x))
    else:
        call("mf_flow_affine_layer_bwd", ptr(image), spec.d, spec.L, order, ptr(x), n, ptr(gy), ptr(glogp), ptr(gx),
             ptr(gslab), rows, int(accumulate), ptr(scratch), scratch.numel(), stream_ptr(x))


def _bwd_plan(spec: FlowSpec, n: int, whole: bool = False):
    """(chunk, scratch_floats, slab_rows) of a backward pass over n particles: the fused kernels need no scratch and
    take the whole batch in one launch per layer; the two-kernel path is chunked to bound its hand-off scratch.  Every
    chunk of a pass must write the same number of slab rows (each workgroup accumulates into its own row), so a ragged
    last chunk is only allowed when it does."""
    lib = _lib.get_lib()
    if spec.wide:
        def need(m):
            return lib.mf_flow_wide_bwd_scratch_floats(m, spec.d, spec.L, spec.bins if spec.kind == "rqs" else 0)

        def rows(m):
            return lib.mf_flow_wide_bwd_slab_rows(m)
    elif spec.kind == "rqs":
        orders = spec.orders if spec.sparse else [None]

        def need(m):
            return max(lib.mf_flow_bwd_scratch_floats(m, spec.d, spec.L, o) for o in orders)

        def rows(m):
            r = {lib.mf_flow_bwd_slab_rows(m, spec.d, spec.L, o) for o in orders}
            if len(r) != 1:
                raise RuntimeError("layers of one flow disagree on the backward variant")
            return r.pop()
    else:
        def need(m):
            return lib.mf_flow_affine_bwd_scratch_floats(m, spec.L)

        def rows(m):
            return lib.mf_flow_affine_bwd_slab_rows(m)

    # whole: one chunk whatever the scratch costs (the wide family's saved activations cover the batch as one tile sequence)
    chunk = n if (need(n) == 0 or whole) else min(n, spec.bwd_chunk)
    return chunk, need(chunk), rows


def pack_images(spec: FlowSpec, flat: torch.Tensor) -> torch.Tensor:
    images = torch.empty(spec.T * spec.image_floats, dtype=_F32, device=flat.device)
    call("mf_gather_f32", ptr(flat), ptr(spec.image_index), ptr(images), images.numel(), 0, stream_ptr(flat))
    return images.view(spec.T, spec.image_floats)


class FlowSampleFn(torch.autograd.Function):
    """(z[N,d], flat parameters) -> (x[N,d], log_prob[N]) through all T autoregressive layers.

    Replaces zuko ``NormalizingFlow.rsample_and_log_prob`` as called by
    mentflow/generate/flows/zuko.py:24-26 (base draw z injected) and its autograd backward.
    Saved for backward: the T layer inputs (N x d each), the packed images and — FlowSpec.act_level — the conditioner's
    hidden tiles / outputs of every layer, written by the forward kernels in the register layout of the fused backward
    (the eager reference keeps every activation for autograd; level 0 recomputes them in the backward instead).

    Two ways to receive the parameter gradient: (a) ``flat`` is an autograd tensor (e.g. ``torch.cat`` of the parameters):
    its gradient is returned to autograd; (b) ``flat`` is a plain buffer, ``trigger`` a leaf that requires grad (so that
    autograd calls this backward) and ``grad_sink(gflat)`` deposits the gradient (AutoregressiveFlow: one flat copy
    into the parameters' .grad views instead of one accumulation kernel per parameter).  In form (b) the parameters are
    NOT part of the autograd graph (see the AutoregressiveFlow docstring for what that implies).  dL/dz is returned when
    ``z`` requires grad (one more layer-0 input gradient + the base-density term -z dL/dlog_prob)."""

    @staticmethod
    def forward(ctx, z: torch.Tensor, flat: torch.Tensor, spec: FlowSpec, grad_reduce=None, trigger=None, grad_sink=None):
        z = _f32c(z)
        flat = _f32c(flat.detach())
        n = z.shape[0]
        images = pack_images(spec, flat)
        logp = torch.empty(n, dtype=_F32, device=z.device)
        level = spec.resolve_act_level(n, z.device)
        act = None
        if level > 0 and spec.wide:
            act = torch.empty(spec.T, _lib.get_lib().mf_flow_wide_act_floats(n, spec.d, spec.L, spec.bins if spec.kind == "rqs" else 0),
                              dtype=_F32, device=z.device)
        elif level > 0:
            act = torch.empty(spec.T, _lib.get_lib().mf_flow_rqs_act_floats(n, spec.d, spec.L, spec.bins, level), dtype=_F32, device=z.device)
        xs = [z]
        for t in range(spec.T):
            y = torch.empty_like(z)
            _layer_fwd(spec, t, images[t], xs[-1], y, logp, logp, t == 0, None if act is None else act[t], level)
            xs.append(y)
        ctx.spec = spec
        ctx.grad_reduce = grad_reduce
        ctx.grad_sink = grad_sink
        ctx.act_level = level
        # a SAVED tensor, so that autograd releases it with the graph right after backward (kept as a plain attribute it lived as
        # long as the loss tensor of the step: two 18 GB buffers alive at C4, and no room for the 16 M-particle batch)
        ctx.save_for_backward(images, act if act is not None else images.new_empty(0), *xs[:-1])
        return xs[-1], logp

    @staticmethod
    def backward(ctx, gx: Optional[torch.Tensor], glogp: Optional[torch.Tensor]):
        spec: FlowSpec = ctx.spec
        images, act, *xs = ctx.saved_tensors
        n = xs[0].shape[0]
        dev = images.device
        gx = torch.zeros(n, spec.d, dtype=_F32, device=dev) if gx is None else _f32c(gx)
        glogp = torch.zeros(n, dtype=_F32, device=dev) if glogp is None else _f32c(glogp)
        level = ctx.act_level
        chunk, scratch_floats, rows_of = _bwd_plan(spec, n, whole=spec.wide and level > 0)
        act = act if level > 0 else None
        if level > 0 and chunk != n:
            raise RuntimeError("the backward variant changed between forward and backward: the saved activations belong to the "
                               "fused kernel (mf_flow_set_bwd_variant)")
        scratch = torch.empty(max(scratch_floats, 1), dtype=_F32, device=dev)
        # chunks grouped by the number of slab rows they write (at most two groups: full chunks and a ragged last one)
        spans = [(a, min(n, a + chunk)) for a in range(0, n, chunk)]
        groups = {}
        for a, b in spans:
            groups.setdefault(rows_of(b - a), []).append((a, b))
        gflat = None
        g = gx
        slabs = {r: torch.empty(spec.T, r, spec.grad_floats, dtype=_F32, device=dev) for r in groups}
        need_gz = ctx.needs_input_grad[0]             # dL/dz: the reference's transform is differentiable in the base draw
        for t in reversed(range(spec.T)):
            gprev = torch.empty_like(g) if (t > 0 or need_gz) else None
            for r, members in groups.items():
                for k, (a, b) in enumerate(members):
                    _layer_bwd(spec, t, images[t], xs[t][a:b], g[a:b], glogp[a:b], None if gprev is None else gprev[a:b],
                               slabs[r][t], k > 0, scratch, None if act is None else act[t], level)
            g = gprev
        for r, slab in slabs.items():
            part = torch.empty(spec.grad_index.numel(), dtype=_F32, device=dev)
            call("mf_flow_grad_reduce", ptr(slab), spec.T, r, spec.grad_floats, ptr(spec.grad_index), ptr(part),
                 part.numel(), stream_ptr(part))
            gflat = part if gflat is None else gflat + part
        # g is now dL/dz through x and the log-det; log_prob also holds the base density logN(z) = -|z|^2/2 - const
        gz = g - xs[0] * glogp[:, None] if need_gz else None
        if ctx.grad_reduce is not None:
            ctx.grad_reduce(gflat)
        if ctx.grad_sink is not None:
            ctx.grad_sink(gflat)
            return gz, None, None, None, None, None
        return gz, gflat, None, None, None, None


def flow_layers_forward(z: torch.Tensor, flat: torch.Tensor, spec: FlowSpec) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """No-grad helper: every intermediate [z, x_1, ..., x_T] and log_prob (forward_steps / sample)."""
    z = _f32c(z)
    images = pack_images(spec, _f32c(flat.detach()))
    logp = torch.empty(z.shape[0], dtype=_F32, device=z.device)
    xs = [z]
    for t in range(spec.T):
        y = torch.empty_like(z)
        _layer_fwd(spec, t, images[t], xs[-1], y, logp, logp, t == 0)
        xs.append(y)
    return xs, logp


def flow_layers_inverse(x: torch.Tensor, flat: torch.Tensor, spec: FlowSpec) -> List[torch.Tensor]:
    """No-grad: [x, T_T^-1(x), ..., z] — the layers in reverse, d autoregressive passes each
    (mentflow/generate/flows/zuko.py:31-32,43-50)."""
    x = _f32c(x)
    images = pack_images(spec, _f32c(flat.detach()))
    zs = [x]
    for t in reversed(range(spec.T)):
        out = torch.empty_like(x)
        if spec.wide:
            call("mf_flow_wide_layer_inv", ptr(images[t]), spec.d, spec.hidden, spec.L, spec.bins if spec.kind == "rqs" else 0,
                 spec.orders[t], ptr(zs[-1]), x.shape[0], ptr(out), stream_ptr(x))
        elif spec.kind == "rqs":
            call("mf_flow_rqs_layer_inv", ptr(images[t]), spec.d, spec.L, spec.bins, spec.orders[t], ptr(zs[-1]),
                 x.shape[0], ptr(out), stream_ptr(x))
        else:
            call("mf_flow_affine_layer_inv", ptr(images[t]), spec.d, spec.L, spec.orders[t], ptr(zs[-1]), x.shape[0],
                 ptr(out), stream_ptr(x))
        zs.append(out)
    return zs


# ------------------------------------------------------------------------------------------------ projections + KDE
class ProjKde1dFn(torch.autograd.Function):
    """x[N,d], V[P,d] -> S[P,B] = sum_n exp(-((x_n.V_p - c_k)/sigma)^2 / 2)   (raw kernel sums).

    Replaces the Python loop of mentflow/simulate/simulate.py:30-33 over LinearTransform.forward
    (simulate/transform.py:67-68) + Histogram1D.project (diagnostics/diagnostics.py:116-122) + the kernel matrix of
    marginal_pdf (diagnostics/histogram.py:37-39)."""

    @staticmethod
    def forward(ctx, x, V, coords, sigma: float, radius: int):
        x, V, coords = _f32c(x), _f32c(V), _f32c(coords)
        P, B = V.shape[0], coords.numel()
        S = torch.empty(P, B, dtype=_F32, device=x.device)
        ws = torch.empty(_lib.get_lib().mf_proj_kde_ws_bytes(P, B) // 8, dtype=torch.int64, device=x.device)
        call("mf_proj_kde1d_fwd", ptr(x), x.shape[0], x.shape[1], ptr(V), P, ptr(coords), B, float(sigma), int(radius),
             ptr(S), ptr(ws), stream_ptr(x))
        ctx.save_for_backward(x, V, coords)
        ctx.sigma, ctx.radius = float(sigma), int(radius)
        return S

    @staticmethod
    def backward(ctx, gS):
        x, V, coords = ctx.saved_tensors
        gx = torch.empty_like(x)
        call("mf_proj_kde1d_bwd", ptr(x), x.shape[0], x.shape[1], ptr(V), V.shape[0], ptr(coords), coords.numel(),
             ctx.sigma, ctx.radius, ptr(_f32c(gS)), ptr(gx), 0, stream_ptr(x))
        return gx, None, None, None, None


class ProjKde2dFn(torch.autograd.Function):
    """x[N,d], V0[P,d], V1[P,d] -> S[P,Bx,By] = sum_n Kx_na Ky_nb  (histogram.py:89-101 / joint_pdf :69)."""

    @staticmethod
    def forward(ctx, x, V0, V1, coords_x, coords_y, sigma_x: float, sigma_y: float, radius_x: int, radius_y: int):
        x, V0, V1, cx, cy = _f32c(x), _f32c(V0), _f32c(V1), _f32c(coords_x), _f32c(coords_y)
        P, Bx, By = V0.shape[0], cx.numel(), cy.numel()
        S = torch.empty(P, Bx, By, dtype=_F32, device=x.device)
        ws = torch.empty(_lib.get_lib().mf_proj_kde_ws_bytes(P, Bx * By) // 8, dtype=torch.int64, device=x.device)
        call("mf_proj_kde2d_fwd", ptr(x), x.shape[0], x.shape[1], ptr(V0), ptr(V1), P, ptr(cx), Bx, float(sigma_x),
             int(radius_x), ptr(cy), By, float(sigma_y), int(radius_y), ptr(S), ptr(ws), stream_ptr(x))
        ctx.save_for_backward(x, V0, V1, cx, cy)
        ctx.args = (float(sigma_x), float(sigma_y), int(radius_x), int(radius_y))
        return S

    @staticmethod
    def backward(ctx, gS):
        x, V0, V1, cx, cy = ctx.saved_tensors
        sx, sy, rx, ry = ctx.args
        gx = torch.empty_like(x)
        call("mf_proj_kde2d_bwd", ptr(x), x.shape[0], x.shape[1], ptr(V0), ptr(V1), V0.shape[0], ptr(cx), cx.numel(), sx,
             rx, ptr(cy), cy.numel(), sy, ry, ptr(_f32c(gS)), ptr(gx), 0, stream_ptr(x))
        return gx, None, None, None, None, None, None, None, None


class MultipoleKickFn(torch.autograd.Function):
    """u = MultipoleTransform(order, strength, skew)(x)  (mentflow/simulate/transform.py:98-143)."""

    @staticmethod
    def forward(ctx, x, order: int, k: float, skew: bool):
        x = _f32c(x)
        u = torch.empty_like(x)
        call("mf_multipole_kick_fwd", ptr(x), x.shape[0], x.shape[1], int(order), float(k), int(bool(skew)), ptr(u),
             stream_ptr(x))
        ctx.save_for_backward(x)
        ctx.args = (int(order), float(k), int(bool(skew)))
        return u

    @staticmethod
    def backward(ctx, gu):
        (x,) = ctx.saved_tensors
        order, k, skew = ctx.args
        gx = torch.empty_like(x)
        call("mf_multipole_kick_bwd", ptr(x), x.shape[0], x.shape[1], order, k, skew, ptr(_f32c(gu)), ptr(gx), stream_ptr(x))
        return gx, None, None, None


DISCREPANCY_KINDS = {"kld": 0, "mae": 1, "mse": 2}


class HistNormDiscFn(torch.autograd.Function):
    """S[P,bins] (+ meas[P,bins]) -> (ghat[P,bins], D[P]).

    normalize: marginal_pdf / joint_pdf normalisation (histogram.py:39-43, :69-73);
    discrepancy: mentflow/loss.py:7-17.  With meas=None only ghat is produced (D is an empty tensor)."""

    @staticmethod
    def forward(ctx, S, meas, normalize: bool, pre_scale: float, cell: float, eps: float, kind: int, pad: float,
                batch_div: float):
        S = _f32c(S)
        P = S.shape[0]
        bins = S[0].numel()
        meas_c = None if meas is None else _f32c(meas)
        ghat = torch.empty_like(S)
        D = torch.empty(P if meas is not None else 0, dtype=_F32, device=S.device)
        call("mf_hist_norm_discrepancy_fwd", ptr(S), P, bins, int(normalize), float(pre_scale), float(cell), float(eps),
             ptr(meas_c), int(kind), float(pad), float(batch_div), ptr(ghat), ptr(D) if meas is not None else None,
             stream_ptr(S))
        ctx.save_for_backward(S, meas_c) if meas is not None else ctx.save_for_backward(S)
        ctx.has_meas = meas is not None
        ctx.args = (int(normalize), float(pre_scale), float(cell), float(eps), int(kind), float(pad), float(batch_div))
        return ghat, D

    @staticmethod
    def backward(ctx, gghat, gD):
        if ctx.has_meas:
            S, meas = ctx.saved_tensors
        else:
            (S,) = ctx.saved_tensors
            meas, gD = None, None
        normalize, pre_scale, cell, eps, kind, pad, batch_div = ctx.args
        gS = torch.empty_like(S)
        gD_c = None if gD is None else _f32c(gD)
        gg_c = None if gghat is None else _f32c(gghat)
        call("mf_hist_norm_discrepancy_bwd", ptr(S), S.shape[0], S[0].numel(), normalize, pre_scale, cell, eps, ptr(meas),
             kind, pad, batch_div, ptr(gD_c), ptr(gg_c), ptr(gS), stream_ptr(S))
        return gS, None, None, None, None, None, None, None, None


class EntropySumsFn(torch.autograd.Function):
    """(x[N,d], logp[N]) -> [sum logp, sum |x|^2]  (entropy.py:58-62, prior.py:25-26)."""

    @staticmethod
    def forward(ctx, x, logp):
        x, logp = _f32c(x), _f32c(logp)
        out = torch.empty(2, dtype=_F32, device=x.device)
        scratch = torch.empty(2048, dtype=torch.float64, device=x.device)      # MF_ENTROPY_SCRATCH_DOUBLES
        call("mf_mc_entropy_sums", ptr(x), ptr(logp), x.shape[0], x.shape[1], ptr(out), ptr(scratch), stream_ptr(x))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        gout = _f32c(gout)
        gx = torch.empty_like(x)
        call("mf_scale_rows", ptr(x), x.shape[0], x.shape[1], ptr(gout[1:2]), 2.0, ptr(gx), 0, stream_ptr(x))
        glogp = gout[0].expand(x.shape[0]).contiguous()
        return gx, glogp


def proj_hist_counts_1d(x, V, edges) -> torch.Tensor:
    x, V, edges = _f32c(x), _f32c(V), _f32c(edges)
    P, B = V.shape[0], edges.numel() - 1
    counts = torch.empty(P, B, dtype=torch.int32, device=x.device)
    call("mf_proj_hist1d_counts", ptr(x), x.shape[0], x.shape[1], ptr(V), P, ptr(edges), B, ptr(counts), stream_ptr(x))
    return counts


def proj_hist_counts_2d(x, V0, V1, edges_x, edges_y) -> torch.Tensor:
    x, V0, V1, ex, ey = _f32c(x), _f32c(V0), _f32c(V1), _f32c(edges_x), _f32c(edges_y)
    P, Bx, By = V0.shape[0], ex.numel() - 1, ey.numel() - 1
    counts = torch.empty(P, Bx, By, dtype=torch.int32, device=x.device)
    call("mf_proj_hist2d_counts", ptr(x), x.shape[0], x.shape[1], ptr(V0), ptr(V1), P, ptr(ex), Bx, ptr(ey), By,
         ptr(counts), stream_ptr(x))
    return counts
