"""ORACLE (test infrastructure) — runs the CPU restatement on the SAME problem a mentflow_amd harness built.

Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Takes plain tensors (weights,
matrices, edges, measurements) out of a ``mentflow_amd.harness.Problem`` and evaluates one MENTFlow.loss() + backward
with the eager dense reference semantics (oracle.model.train_step_loss)."""
from __future__ import annotations

import torch

from . import flow as of
from . import model as om


def flow_spec_from_generator(generator, dtype=torch.float32) -> of.FlowSpec:
    spec = of.FlowSpec(generator.features, generator.kind, generator.bins)
    for layer in generator.layers:
        lins = layer.linears()
        spec.layers.append(of.ARLayer(layer.order.cpu(),
                                      [l.weight.detach().cpu().to(dtype).clone() for l in lins],
                                      [l.bias.detach().cpu().to(dtype).clone() for l in lins],
                                      [l.mask.cpu() for l in lins]))
    return spec


def oracle_problem(prob, dtype=torch.float32):
    spec = flow_spec_from_generator(prob.model.generator, dtype)
    def restate(t):
        kind = type(t).__name__
        if kind == "LinearTransform":
            return om.LinearTransform(t.matrix.detach().cpu().to(dtype))
        if kind == "MultipoleTransform":
            return om.MultipoleTransform(t.order, t.strength, t.skew)
        if kind == "CompositeTransform":
            return om.CompositeTransform(*[restate(c) for c in t.transforms])
        raise TypeError(kind)

    transforms = [restate(t) for t in prob.transforms]
    d0 = prob.diagnostics[0][0]
    if d0.ndim == 1:
        diag = om.Histogram1D(edges=d0.edges.cpu().to(dtype), bandwidth=d0.bandwidth_bins, axis=d0.axis)
    else:
        diag = om.Histogram2D(axis=d0.axis, edges=(d0.edges_x.cpu().to(dtype), d0.edges_y.cpu().to(dtype)),
                              bandwidth=d0.bandwidth_bins)
    diagnostics = [[diag] for _ in transforms]
    measurements = [[m.detach().cpu().to(dtype) for m in row] for row in prob.measurements]
    prior = om.GaussianPrior(prob.cfg["ndim"], prob.cfg["prior_scale"], dtype=dtype)
    disc = {"kld": om.kl_divergence, "mae": om.mean_absolute_error, "mse": om.mean_square_error}[prob.cfg["discrepancy"]]
    return spec, transforms, diagnostics, measurements, prior, disc


def oracle_step(prob, z: torch.Tensor, dtype=torch.float32, backward: bool = True):
    """(L, H, [D_p], flat parameter gradient) of one training step with the base draw z injected."""
    spec, transforms, diagnostics, measurements, prior, disc = oracle_problem(prob, dtype)
    params = spec.parameters()
    for p in params:
        p.requires_grad_(backward)
    L, H, D, x, logp = om.train_step_loss(z.detach().cpu().to(dtype), spec, transforms, diagnostics, measurements, prior,
                                          float(prob.model.penalty_parameter), disc)
    g = None
    if backward:
        L.backward()
        g = torch.cat([p.grad.reshape(-1) for p in params])
    return L.detach(), H.detach(), [d.detach() for d in D], g
