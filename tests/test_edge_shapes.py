"""Edge shapes on the kernels (emulated in the CPU suite, the real library with -m gpu): ragged and tiny batches around
the 32-particle tile and the 4-tile workgroup group of the fused backward, every supported d / bins / hidden_layers
combination, odd projection / bin counts and d = 1..8 for the projection kernels — against the oracle."""
import pytest
import torch

from conftest import set_bwd_variant

import mentflow_amd as mf
from mentflow_amd import ops
from oracle import flow as of
from oracle.harness import flow_spec_from_generator


def _gen(backend, d, bins, L, kind="nsf", seed=0, steep=True):
    """steep: last conditioner layer x 3 + random biases (slopes far from 1: exercises every branch of the spline, at the
    price of amplifying fp32 rounding: the loose gates); steep=False: the default initialisation the reference trains from
    (tight gates)."""
    torch.manual_seed(seed)
    kws = dict(input_features=d, output_features=d, hidden_layers=L, hidden_units=64, transforms=2)
    if kind == "nsf":
        kws["bins"] = bins
    g = mf.generate.build_generator(kind, **kws)
    if steep:
        with torch.no_grad():
            for layer in g.layers:
                lin = layer.linears()[-1]
                lin.weight.mul_(3.0)
                lin.bias.add_(0.5 * torch.randn_like(lin.bias))
    return g.to(backend)


def _check_default_init(gen, backend, n, d, seed):
    """Default initialisation, the tight gates of tests/test_flow_kernels.py::test_nsf_default_init_tight_gates: x 1e-5,
    log_prob 1e-4, parameter gradients 5e-4 of the largest entry (fp64 oracle)."""
    torch.manual_seed(seed)
    z = torch.randn(n, d)
    wx, wl = torch.randn(n, d), torch.randn(n)
    x, lp = gen.sample_and_log_prob(n, z=z.to(backend))
    ((x * wx.to(backend)).sum() + (lp * wl.to(backend)).sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu().double()
    s64 = flow_spec_from_generator(gen, torch.float64)
    ps = s64.parameters()
    for p in ps:
        p.requires_grad_(True)
    xo, lo = of.sample_and_log_prob(z.double(), s64)
    ((xo * wx.double()).sum() + (lo * wl.double()).sum()).backward()
    go = torch.cat([p.grad.reshape(-1) for p in ps])
    ex, el = float((x.detach().cpu() - xo).abs().max()), float((lp.detach().cpu() - lo).abs().max())
    eg = float((g - go).abs().max() / go.abs().max())
    assert ex < 1e-5 and el < 1e-4 and eg < 5e-4, (ex, el, eg)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 257])
@pytest.mark.parametrize("variant", ["fused", "two-kernel"])
def test_flow_ragged_batches_default_init_tight_gates(backend, n, variant, monkeypatch):
    set_bwd_variant(monkeypatch, "1" if variant == "fused" else "0")
    _check_default_init(_gen(backend, 6, 20, 3, steep=False), backend, n, 6, seed=n)


@pytest.mark.parametrize("d,bins,L", [(2, 8, 2), (3, 20, 2), (4, 8, 3), (5, 20, 3), (6, 8, 2), (7, 8, 3), (7, 20, 2), (6, 12, 3)])
def test_flow_every_instance_default_init_tight_gates(backend, d, bins, L):
    _check_default_init(_gen(backend, d, bins, L, seed=d, steep=False), backend, 70, d, seed=3)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 257])
@pytest.mark.parametrize("variant", ["fused", "two-kernel"])
def test_flow_ragged_batches(backend, n, variant, monkeypatch):
    set_bwd_variant(monkeypatch, "1" if variant == "fused" else "0")
    gen = _gen(backend, 6, 20, 3)
    torch.manual_seed(n)
    z = torch.randn(n, 6) * 1.4
    wx, wl = torch.randn(n, 6), torch.randn(n)
    x, lp = gen.sample_and_log_prob(n, z=z.to(backend))
    ((x * wx.to(backend)).sum() + (lp * wl.to(backend)).sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu().double()
    s64 = flow_spec_from_generator(gen, torch.float64)
    ps = s64.parameters()
    for p in ps:
        p.requires_grad_(True)
    xo, lo = of.sample_and_log_prob(z.double(), s64)
    ((xo * wx.double()).sum() + (lo * wl.double()).sum()).backward()
    go = torch.cat([p.grad.reshape(-1) for p in ps])
    assert (x.detach().cpu() - xo).abs().max() < 5e-5 and (lp.detach().cpu() - lo).abs().max() < 5e-4
    assert (g - go).abs().max() < 2e-3 * go.abs().max()
    zb = gen.inverse(x.detach())
    with torch.no_grad():
        xr, _ = gen.sample_and_log_prob(n, z=zb)
    assert (xr.cpu() - x.detach().cpu()).abs().max() < 5e-5 * max(1.0, float(x.detach().abs().max()))


@pytest.mark.parametrize("d,bins,L", [(2, 8, 2), (3, 20, 2), (4, 8, 3), (5, 20, 3), (6, 8, 2), (7, 8, 3), (7, 20, 2)])
def test_flow_every_compiled_instance(backend, d, bins, L):
    gen = _gen(backend, d, bins, L, seed=d)
    n = 70
    torch.manual_seed(3)
    z = torch.randn(n, d) * 1.3
    wx, wl = torch.randn(n, d), torch.randn(n)
    x, lp = gen.sample_and_log_prob(n, z=z.to(backend))
    ((x * wx.to(backend)).sum() + (lp * wl.to(backend)).sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu().double()
    s64 = flow_spec_from_generator(gen, torch.float64)
    ps = s64.parameters()
    for p in ps:
        p.requires_grad_(True)
    xo, lo = of.sample_and_log_prob(z.double(), s64)
    ((xo * wx.double()).sum() + (lo * wl.double()).sum()).backward()
    go = torch.cat([p.grad.reshape(-1) for p in ps])
    assert (x.detach().cpu() - xo).abs().max() < 5e-5 and (lp.detach().cpu() - lo).abs().max() < 5e-4
    assert (g - go).abs().max() < 2e-3 * go.abs().max()


def test_unsupported_shapes_fail_loudly(backend):
    """hidden_units above 128 and mixed widths are refused when the generator is built (65 .. 128 run the wide family:
    tests/test_flow_wide.py; narrower conditioners run zero-padded: test_narrow_conditioner_matches_oracle); bins beyond the 32 slots of a lane half and
    hidden_layers without a compiled instance are refused with a message that names what exists — nothing falls back
    silently."""
    with pytest.raises(NotImplementedError):
        mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=129, transforms=1,
                                    bins=8)
    gen = mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=64,
                                      transforms=1, bins=22).to(backend)
    with pytest.raises(NotImplementedError, match="2 <= bins <= 21"):
        gen.sample_and_log_prob(8, z=torch.randn(8, 2).to(backend))
    gen = mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=5, hidden_units=64,
                                      transforms=1, bins=8).to(backend)
    with pytest.raises(RuntimeError, match="no RQS kernel instance"):
        gen.sample_and_log_prob(8, z=torch.randn(8, 2).to(backend))


@pytest.mark.parametrize("d,bins,L", [(2, 2, 2), (3, 5, 3), (6, 12, 3), (4, 16, 2), (6, 21, 3), (7, 13, 2)])
@pytest.mark.parametrize("variant", ["fused", "two-kernel"])
def test_flow_run_time_bins_instance(backend, d, bins, L, variant, monkeypatch):
    """bins outside the compile-time instances (8, 20) run through the run-time-bins kernels (slots laid out for 21 bins):
    forward, both backward variants (d = 7 always takes the two-kernel path) and the inverse against the oracle, gates as
    for the compiled instances."""
    set_bwd_variant(monkeypatch, "1" if variant == "fused" else "0")
    gen = _gen(backend, d, bins, L, seed=bins)
    n = 70
    torch.manual_seed(bins)
    z = torch.randn(n, d) * 1.3
    wx, wl = torch.randn(n, d), torch.randn(n)
    x, lp = gen.sample_and_log_prob(n, z=z.to(backend))
    ((x * wx.to(backend)).sum() + (lp * wl.to(backend)).sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu().double()
    s64 = flow_spec_from_generator(gen, torch.float64)
    ps = s64.parameters()
    for p in ps:
        p.requires_grad_(True)
    xo, lo = of.sample_and_log_prob(z.double(), s64)
    ((xo * wx.double()).sum() + (lo * wl.double()).sum()).backward()
    go = torch.cat([p.grad.reshape(-1) for p in ps])
    assert (x.detach().cpu() - xo).abs().max() < 5e-5 and (lp.detach().cpu() - lo).abs().max() < 5e-4
    assert (g - go).abs().max() < 2e-3 * go.abs().max()
    zb = gen.inverse(x.detach())
    assert (zb.cpu() - z).abs().max() < 2e-4
    lpx = gen.log_prob(x.detach())
    assert (lpx.cpu() - lp.detach().cpu()).abs().max() < 2e-3


@pytest.mark.parametrize("d,P,B,n", [(1, 1, 2, 5), (2, 3, 7, 63), (3, 17, 33, 65), (5, 101, 16, 300), (8, 4, 128, 513)])
def test_kde1d_odd_shapes(backend, d, P, B, n):
    torch.manual_seed(d * 100 + P)
    x = (torch.randn(n, d) * 1.2)
    V = torch.randn(P, d)
    V = V / V.norm(dim=1, keepdim=True)
    edges = torch.linspace(-3.0, 3.0, B + 1)
    coords = 0.5 * (edges[1:] + edges[:-1])
    delta = float(edges[1] - edges[0])
    gS = torch.randn(P, B)
    xs = x.to(backend).clone().requires_grad_(True)
    S = ops.ProjKde1dFn.apply(xs, V.to(backend), coords.to(backend), 0.5 * delta, ops.kde_radius(0.5))
    (S * gS.to(backend)).sum().backward()
    xo = x.double().clone().requires_grad_(True)
    u = xo @ V.double().T                                                  # [n, P]
    K = torch.exp(-0.5 * ((u[:, :, None] - coords.double()[None, None, :]) / (0.5 * delta)) ** 2)
    So = K.sum(0)
    (So * gS.double()).sum().backward()
    torch.testing.assert_close(S.detach().cpu().double(), So.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xs.grad.cpu().double(), xo.grad, rtol=1e-3, atol=2e-5 * float(xo.grad.abs().max()) + 1e-6)


@pytest.mark.parametrize("d,P,Bx,By,n", [(2, 1, 5, 9, 40), (4, 3, 16, 11, 130), (6, 2, 85, 40, 257)])
def test_kde2d_odd_shapes(backend, d, P, Bx, By, n):
    torch.manual_seed(d * 10 + P)
    x = torch.randn(n, d)
    V0 = torch.randn(P, d)
    V1 = torch.randn(P, d)
    ex, ey = torch.linspace(-3.0, 3.0, Bx + 1), torch.linspace(-2.5, 2.5, By + 1)
    cx, cy = 0.5 * (ex[1:] + ex[:-1]), 0.5 * (ey[1:] + ey[:-1])
    sx, sy = 0.5 * float(ex[1] - ex[0]), 0.5 * float(ey[1] - ey[0])
    gS = torch.randn(P, Bx, By)
    xs = x.to(backend).clone().requires_grad_(True)
    S = ops.ProjKde2dFn.apply(xs, V0.to(backend), V1.to(backend), cx.to(backend), cy.to(backend), sx, sy, 4, 4)
    (S * gS.to(backend)).sum().backward()
    xo = x.double().clone().requires_grad_(True)
    u0, u1 = xo @ V0.double().T, xo @ V1.double().T
    Kx = torch.exp(-0.5 * ((u0[:, :, None] - cx.double()) / sx) ** 2)     # [n, P, Bx]
    Ky = torch.exp(-0.5 * ((u1[:, :, None] - cy.double()) / sy) ** 2)
    So = torch.einsum("npa,npb->pab", Kx, Ky)
    (So * gS.double()).sum().backward()
    torch.testing.assert_close(S.detach().cpu().double(), So.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xs.grad.cpu().double(), xo.grad, rtol=1e-3, atol=2e-5 * float(xo.grad.abs().max()) + 1e-6)


@pytest.mark.parametrize("bw", [0.7, 0.12])
def test_kde_radius_4_with_a_bandwidth_outside_the_specialised_range(backend, bw):
    """The C ABI takes radius and sigma independently (include/mentflow_hip.h): radius 4 with s = delta / sigma far from
    [2, 2.6] must not take the factorised window / dead-corner skip (s = 1.43: the window TRUNCATES, the result is the
    sum over the 9 bins given; s = 8.3: the factorised tail would overflow to inf * 0 = NaN).  Reference: the dense sum
    restricted to the same window (fp64)."""
    torch.manual_seed(5)
    n, d, P, B = 400, 3, 4, 24
    x = torch.randn(n, d) * 1.1
    V = torch.randn(P, d)
    V = V / V.norm(dim=1, keepdim=True)
    edges = torch.linspace(-3.0, 3.0, B + 1)
    coords = 0.5 * (edges[1:] + edges[:-1])
    delta = float(edges[1] - edges[0])
    sigma = bw * delta
    gS = torch.randn(P, B)

    def windowed(u, c, sig):                              # [n, P, B] kernel values of the bins within 4 of the centre bin
        kc = torch.round((u - c[0]) / delta)
        k = torch.arange(c.numel(), dtype=u.dtype)
        inside = (k[None, None, :] - kc[:, :, None]).abs() <= 4
        return torch.exp(-0.5 * ((u[:, :, None] - c[None, None, :]) / sig) ** 2) * inside

    xs = x.to(backend).clone().requires_grad_(True)
    S = ops.ProjKde1dFn.apply(xs, V.to(backend), coords.to(backend), sigma, 4)
    (S * gS.to(backend)).sum().backward()
    assert torch.isfinite(S).all() and torch.isfinite(xs.grad).all()
    xo = x.double().clone().requires_grad_(True)
    So = windowed(xo @ V.double().T, coords.double(), sigma).sum(0)
    (So * gS.double()).sum().backward()
    torch.testing.assert_close(S.detach().cpu().double(), So.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xs.grad.cpu().double(), xo.grad, rtol=1e-3, atol=2e-5 * float(xo.grad.abs().max()) + 1e-6)
    # 2-D, both axes at that bandwidth
    gS2 = torch.randn(P, B, B)
    V1 = torch.randn(P, d)
    xs = x.to(backend).clone().requires_grad_(True)
    S2 = ops.ProjKde2dFn.apply(xs, V.to(backend), V1.to(backend), coords.to(backend), coords.to(backend), sigma, sigma, 4, 4)
    (S2 * gS2.to(backend)).sum().backward()
    assert torch.isfinite(S2).all() and torch.isfinite(xs.grad).all()
    xo = x.double().clone().requires_grad_(True)
    Kx = windowed(xo @ V.double().T, coords.double(), sigma)
    Ky = windowed(xo @ V1.double().T, coords.double(), sigma)
    S2o = torch.einsum("npa,npb->pab", Kx, Ky)
    (S2o * gS2.double()).sum().backward()
    torch.testing.assert_close(S2.detach().cpu().double(), S2o.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xs.grad.cpu().double(), xo.grad, rtol=1e-3, atol=2e-5 * float(xo.grad.abs().max()) + 1e-6)
