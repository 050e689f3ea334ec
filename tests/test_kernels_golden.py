"""Parity of the kernels (through the C ABI and the mentflow-compatible Python layer) with the REFERENCE'S OWN
outputs (tests/golden/ref_*.npz) and with the oracle.  Each test runs on the emulated host build ("emu", CPU
container) and on the real gfx950 library ("hip", -m gpu).

Tolerances (fp32, stated per SURVEY.md §8d): histograms rtol 2e-5 + atol 1e-6; H and L rtol 1e-5 (+ atol 1e-6);
dL/dx rtol 1e-3 + atol 1e-7 * max|g|... (each assertion spells its own numbers)."""
import numpy as np
import pytest
import torch

import mentflow_amd as mf
from mentflow_amd import ops
from conftest import load_golden


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.detach().cpu().float(), b.float(), rtol=rtol, atol=atol)


class Injected(mf.generate.GenerativeModel):
    """Generator stub returning fixed (x, log_prob) leaves — same trick the golden generator used."""

    def __init__(self, x, logp):
        super().__init__()
        self.x, self.logp = x, logp
        self._dummy = torch.nn.Parameter(torch.zeros(1))

    def sample_and_log_prob(self, n):
        return self.x, self.logp

    def sample(self, n):
        return self.x


@pytest.mark.parametrize("bins", [64, 85])
def test_kde1d_vs_reference(backend, bins):
    g = load_golden(f"ref_kde1d_B{bins}")
    u = g["u"].to(backend)[:, None].clone().requires_grad_(True)          # N x 1 "phase space"
    diag = mf.diagnostics.Histogram1D(edges=g["edges"], bandwidth=0.5, axis=0).to(backend)
    hist = diag(u)
    (hist * g["w"].to(backend)).sum().backward()
    close(hist, g["hist"], 2e-5, 1e-6)
    close(u.grad[:, 0], g["grad_u"], 1e-3, 2e-6)
    # standalone sanity: normalisation sum(ghat)*delta = 1
    assert abs(float(hist.sum() * diag.resolution) - 1.0) < 1e-5


@pytest.mark.parametrize("bins", [64, 85])
def test_kde2d_vs_reference(backend, bins):
    g = load_golden(f"ref_kde2d_B{bins}")
    u = g["u"].to(backend).clone().requires_grad_(True)
    diag = mf.diagnostics.Histogram2D(axis=(0, 1), edges=(g["edges_x"], g["edges_y"]), bandwidth=(0.5, 0.5)).to(backend)
    hist = diag(u)
    (hist * g["w"].to(backend)).sum().backward()
    close(hist, g["hist"], 2e-5, 1e-7)
    close(u.grad, g["grad_u"], 1e-3, 2e-6)


def test_hard_histograms_vs_reference(backend):
    g = load_golden("ref_hist_hard")
    x = g["x"].to(backend)
    d1 = mf.diagnostics.Histogram1D(edges=g["edges1"], bandwidth=0.5, axis=0, kde=False).to(backend)
    h1 = d1(x)
    close(h1, g["hist1"], 1e-6, 1e-7)
    counts = ops.proj_hist_counts_1d(x, torch.eye(6, device=backend)[:1].contiguous(), g["edges1"].to(backend))
    ref_counts = torch.histogram(g["x"][:, 0], g["edges1"]).hist
    assert torch.equal(counts[0].cpu().float(), ref_counts)              # integer counts: bit exact
    d2 = mf.diagnostics.Histogram2D(axis=(0, 2), edges=(g["edges2x"], g["edges2y"]), bandwidth=(0.5, 0.5),
                                    kde=False).to(backend)
    close(d2(x), g["hist2"], 1e-6, 1e-7)


def test_forward_list_structure(backend):
    g = load_golden("ref_forward_list")
    transforms = []
    for v in g["V"]:
        M = torch.eye(6)
        M[0, :] = v
        transforms.append(mf.simulate.LinearTransform(M).to(backend))
    diag = mf.diagnostics.Histogram1D(edges=g["edges"], bandwidth=0.5, axis=0).to(backend)
    preds = mf.simulate.forward(g["x"].to(backend), transforms, [[diag] for _ in transforms])
    assert len(preds) == 25 and all(len(p) == 1 for p in preds)
    close(torch.stack([p[0] for p in preds]), g["preds"], 2e-5, 1e-6)


def test_losses_vs_reference(backend):
    g = load_golden("ref_losses")
    for suffix in ("", "2"):
        pred, targ = g["pred" + suffix].to(backend), g["targ" + suffix].to(backend)
        close(mf.loss.kl_divergence(pred, targ), g["kl" + suffix], 1e-5, 1e-7)
        close(mf.loss.mean_absolute_error(pred, targ), g["mae" + suffix], 1e-5, 1e-7)
        close(mf.loss.mean_square_error(pred, targ), g["mse" + suffix], 1e-5, 1e-7)
    pred = g["pred"].to(backend).clone().requires_grad_(True)
    mf.loss.kl_divergence(pred, g["targ"].to(backend)).backward()
    expect = -g["targ"] / (g["pred"] + 1e-12) / 64
    close(pred.grad, expect, 1e-5, 1e-8)


def test_entropy_vs_reference(backend):
    g = load_golden("ref_entropy_mc")
    x, lp = g["x"].to(backend), g["log_prob"].to(backend)
    for s in (1.0, 3.0):
        est = mf.entropy.MonteCarloEntropyEstimator(prior=mf.prior.Gaussian(6, s))
        close(est(x, lp), g[f"H_scale{s}"], 1e-5, 1e-6)
        close(mf.prior.Gaussian(6, s).log_prob(x), g[f"prior_logp_scale{s}"], 1e-5, 1e-5)
    close(mf.entropy.MonteCarloEntropyEstimator(prior=None)(x, lp), g["H_noprior"], 1e-5, 1e-6)


def _loss_case(backend, g, transforms, diagnostics, ndim):
    meas = [[m.to(backend)] for m in g["measurements"]]
    for mu in (0, 500):
        x = g["x"].to(backend).clone().requires_grad_(True)
        lp = g["log_prob"].to(backend).clone().requires_grad_(True)
        prior = mf.prior.Gaussian(ndim, float(g["prior_scale"]))
        model = mf.MENTFlow(transforms=transforms, diagnostics=diagnostics, measurements=meas,
                            generator=Injected(x, lp), prior=prior,
                            entropy_estimator=mf.entropy.MonteCarloEntropyEstimator(prior=prior),
                            discrepancy_function=mf.loss.kl_divergence, penalty_parameter=float(mu))
        L, H, D = model.loss(x.shape[0])
        assert isinstance(D, list) and len(D) == len(transforms)
        L.backward()
        close(H, g[f"H_mu{mu}"], 1e-5, 1e-6)
        close(torch.stack(D), g[f"D_mu{mu}"], 2e-4, 1e-7)
        # L = H + mu * mean(D): D is a cancellation-prone O(1e-2) difference of O(1) sums, good to ~1e-6 absolute
        # in fp32 on either side, so the absolute tolerance of L carries a mu * 1e-6 term
        close(L, g[f"L_mu{mu}"], 2e-5, 1e-5 + mu * 1e-6)
        gmax = float(g[f"gx_mu{mu}"].abs().max())
        close(x.grad, g[f"gx_mu{mu}"], 1e-3, 2e-5 * gmax)
        close(lp.grad, g[f"glogp_mu{mu}"], 1e-6, 1e-9)
        # Trainer-style use of the triple (train.py:165-177)
        assert not (torch.isinf(L) or torch.isnan(L))
        float(L), float(H), float(sum(D) / len(D))


@pytest.mark.parametrize("P,xmax", [(25, 4.0), (100, 3.5)])
def test_mentflow_loss_nd_1d(backend, P, xmax):
    g = load_golden(f"ref_mentflow_loss_1d_P{P}")
    transforms = [mf.simulate.LinearTransform(M).to(backend) for M in g["matrices"]]
    diag = mf.diagnostics.Histogram1D(edges=torch.linspace(-xmax, xmax, 65), bandwidth=0.5, axis=0).to(backend)
    _loss_case(backend, g, transforms, [[diag] for _ in transforms], 6)


def test_mentflow_loss_2d_rotations(backend):
    g = load_golden("ref_mentflow_loss_2d_P7")
    transforms = [mf.simulate.LinearTransform(M).to(backend) for M in g["matrices"]]
    diag = mf.diagnostics.Histogram1D(edges=torch.linspace(-3.5, 3.5, 86), bandwidth=0.5, axis=0).to(backend)
    _loss_case(backend, g, transforms, [[diag] for _ in transforms], 2)


def test_mentflow_loss_nd_2d_corner(backend):
    g = load_golden("ref_mentflow_loss_nd2d_corner15")
    transforms = [mf.simulate.LinearTransform(M).to(backend) for M in g["matrices"]]
    e = torch.linspace(-3.5, 3.5, 49)
    diag = mf.diagnostics.Histogram2D(axis=(0, 2), edges=(e, e), bandwidth=(0.5, 0.5)).to(backend)
    _loss_case(backend, g, transforms, [[diag] for _ in transforms], 6)


@pytest.mark.parametrize("d", [2, 4, 6])
def test_multipole_kick_vs_reference(backend, d):
    """mf_multipole_kick_fwd/bwd against the reference's MultipoleTransform outputs and autograd gradients.
    fp32: the kernel evaluates z^(order-1) by complex multiplication, the reference by the expanded polynomial, so
    the two differ by rounding only: rtol 1e-5, atol 1e-5 * max|value|."""
    g = load_golden("ref_multipole")
    for order in (3, 4, 5):
        for skew in (False, True):
            tag = f"d{d}_o{order}_s{int(skew)}"
            x = g[f"x_d{d}"].to(backend).clone().requires_grad_(True)
            t = mf.simulate.MultipoleTransform(order, 0.7 * order, skew)
            u = t(x)
            (u * g[f"w_d{d}"].to(backend)).sum().backward()
            close(u, g[f"u_{tag}"], 1e-5, 1e-5 * float(g[f"u_{tag}"].abs().max()))
            close(x.grad, g[f"gx_{tag}"], 1e-5, 1e-5 * float(g[f"gx_{tag}"].abs().max()))
            close(t.inverse(u.detach()), g[f"inv_{tag}"], 1e-5, 1e-5 * float(g[f"inv_{tag}"].abs().max()))
    with pytest.raises(ValueError):
        mf.simulate.MultipoleTransform(2, 1.0)


def test_mentflow_loss_2d_nonlinear(backend):
    """rec_2d/nonlinear (4 x CompositeTransform(multipole, rotation), 85 bins, xmax 4.5) in the flow configuration
    (MC entropy + KL) and the NN configuration (EmptyEntropyEstimator + MAE, log_prob None)."""
    g = load_golden("ref_mentflow_loss_2d_nonlinear")
    rot = mf.simulate.rotation_matrix(np.radians(float(g["angle_deg"]))).type(torch.float32)
    transforms = [mf.simulate.CompositeTransform(mf.simulate.MultipoleTransform(int(g["order"]), float(s)),
                                                 mf.simulate.LinearTransform(rot)).to(backend)
                  for s in g["strengths"]]
    diag = mf.diagnostics.Histogram1D(edges=g["edges"], bandwidth=0.5, axis=0).to(backend)
    meas = [[m.to(backend)] for m in g["measurements"]]
    for tag in ("flow", "nn"):
        x = g["x"].to(backend).clone().requires_grad_(True)
        lp = g["log_prob"].to(backend).clone().requires_grad_(True) if tag == "flow" else None
        prior = mf.prior.Gaussian(2, 1.0)
        est = mf.entropy.MonteCarloEntropyEstimator(prior=prior) if tag == "flow" else mf.entropy.EmptyEntropyEstimator()
        disc = mf.loss.kl_divergence if tag == "flow" else mf.loss.mean_absolute_error
        model = mf.MENTFlow(transforms=transforms, diagnostics=[[diag] for _ in transforms], measurements=meas,
                            generator=Injected(x, lp), prior=prior, entropy_estimator=est, discrepancy_function=disc,
                            penalty_parameter=500.0)
        L, H, D = model.loss(x.shape[0])
        L.backward()
        close(torch.stack(D), g[f"D_{tag}"], 2e-4, 1e-7)
        close(torch.as_tensor(float(H)), g[f"H_{tag}"], 1e-5, 1e-6)
        close(L, g[f"L_{tag}"], 2e-5, 1e-5 + 500 * 1e-6)
        gmax = float(g[f"gx_{tag}"].abs().max())
        close(x.grad, g[f"gx_{tag}"], 1e-3, 2e-5 * gmax)
        if lp is not None:
            close(lp.grad, g[f"glogp_{tag}"], 1e-6, 1e-9)
    # the un-fused list API gives the same predictions
    preds = mf.simulate.forward(g["x"].to(backend), transforms, [[diag] for _ in transforms])
    assert len(preds) == 4 and all(len(p) == 1 and p[0].shape == (85,) for p in preds)


def test_kde1d_backward_nonfinite_rows_same_in_both_window_variants(backend):
    """the unrolled radius-4 backward window and the generic-radius loop must treat inf / NaN / far
    out-of-range projections alike: such a particle touches no bin, its gradient row is exactly 0 — never NaN."""
    torch.manual_seed(3)
    n, d, P, B = 300, 3, 5, 32
    x = torch.randn(n, d)
    x[5, 0] = float("inf")
    x[6, 1] = float("-inf")
    x[7, 2] = float("nan")
    x[8] = 1.0e30
    x[9] = -3.0e38
    V = torch.randn(P, d)
    V = V / V.norm(dim=1, keepdim=True)
    edges = torch.linspace(-4.0, 4.0, B + 1)
    coords = 0.5 * (edges[1:] + edges[:-1])
    delta = float(edges[1] - edges[0])
    gS = torch.randn(P, B)
    bad = [5, 6, 7, 8, 9]
    outs = []
    for bw_bins in (0.5, 0.5 + 1e-3):          # radius 4 (unrolled, RT = 4) and radius 5 (runtime-radius loop)
        R = ops.kde_radius(bw_bins)
        xs = x.to(backend).clone().requires_grad_(True)
        S = ops.ProjKde1dFn.apply(xs, V.to(backend), coords.to(backend), bw_bins * delta, R)
        (S * gS.to(backend)).sum().backward()
        g = xs.grad.cpu()
        assert torch.isfinite(g).all(), f"radius {R}: non-finite gradient rows {torch.nonzero(~torch.isfinite(g))[:4]}"
        assert torch.equal(g[bad], torch.zeros(len(bad), d)), f"radius {R}: {g[bad]}"
        outs.append((R, g))
    assert outs[0][0] == 4 and outs[1][0] == 5
    good = [i for i in range(n) if i not in bad]
    # both variants agree on the ordinary rows (bandwidths differ by 0.2 %: loose tolerance, the point is no NaN leak)
    torch.testing.assert_close(outs[0][1][good], outs[1][1][good], rtol=0.05, atol=0.05 * float(outs[0][1].abs().max()))


@pytest.mark.parametrize("bw", [0.3, 0.45, 0.6, 1.0])
def test_kde_other_bandwidths_vs_oracle(backend, bw):
    """Bandwidths other than the reference default 0.5 bin widths: 0.45 still takes the radius-4 window (exact central
    bins + factorised tail, s = 2.22), 0.3 / 0.6 / 1.0 the run-time-radius loops (radius 3, 5, 9; 2-D: radius <= 5 only).
    Values and gradients against the pinned dense oracle (oracle/kde.py, fp64)."""
    from oracle import kde as okde
    torch.manual_seed(17)
    n, B = 700, 40
    u = torch.randn(n, 2) * 1.3
    u[0, 0], u[1, 1], u[2, 0] = 3.9, -4.4, 7.0                    # near / beyond the histogram range
    edges = torch.linspace(-4.0, 4.0, B + 1)
    w1 = torch.randn(B)
    # 1-D
    x = u[:, :1].clone().to(backend).requires_grad_(True)
    diag = mf.diagnostics.Histogram1D(edges=edges, bandwidth=bw, axis=0).to(backend)
    hist = diag(x)
    (hist * w1.to(backend)).sum().backward()
    xo = u[:, 0].double().clone().requires_grad_(True)
    ho = okde.kde_histogram_1d(xo, edges.double(), bandwidth=bw * float(edges[1] - edges[0]))
    (ho * w1.double()).sum().backward()
    close(hist, ho.float(), 2e-5, 1e-6)
    close(x.grad[:, 0], xo.grad.float(), 1e-3, 2e-6)
    if bw > 0.6:
        return                                                     # the 2-D kernels support radii up to 5 bins
    # 2-D
    w2 = torch.randn(B, B)
    x2 = u.clone().to(backend).requires_grad_(True)
    d2 = mf.diagnostics.Histogram2D(axis=(0, 1), edges=(edges, edges), bandwidth=(bw, 0.5)).to(backend)
    h2 = d2(x2)
    (h2 * w2.to(backend)).sum().backward()
    uo = u.double().clone().requires_grad_(True)
    delta = float(edges[1] - edges[0])
    h2o = okde.kde_histogram_2d(uo[:, 0], uo[:, 1], (edges.double(), edges.double()), bandwidth=(bw * delta, 0.5 * delta))
    (h2o * w2.double()).sum().backward()
    close(h2, h2o.float(), 2e-5, 1e-7)
    close(x2.grad, uo.grad.float(), 1e-3, 2e-6)
