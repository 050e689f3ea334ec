"""Edge shapes on the kernels (emulated in the CPU suite, the real library with -m gpu): ragged and tiny batches around
the 32-particle tile and the 4-tile workgroup group of the fused backward, every supported d / bins / hidden_layers
combination, odd projection / bin counts and d = 1..8 for the projection kernels — against the oracle."""
import pytest
import torch

import mentflow_amd as mf
from mentflow_amd import ops
from oracle import flow as of
from oracle.harness import flow_spec_from_generator


def _gen(backend, d, bins, L, kind="nsf", seed=0):
    torch.manual_seed(seed)
    kws = dict(input_features=d, output_features=d, hidden_layers=L, hidden_units=64, transforms=2)
    if kind == "nsf":
        kws["bins"] = bins
    g = mf.generate.build_generator(kind, **kws)
    with torch.no_grad():
        for layer in g.layers:
            lin = layer.linears()[-1]
            lin.weight.mul_(3.0)
            lin.bias.add_(0.5 * torch.randn_like(lin.bias))
    return g.to(backend)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 127, 128, 129, 257])
@pytest.mark.parametrize("variant", ["fused", "two-kernel"])
def test_flow_ragged_batches(backend, n, variant, monkeypatch):
    monkeypatch.setenv("MENTFLOW_BWD_FUSED", "1" if variant == "fused" else "0")
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
    """hidden_units != 64 is refused when the generator is built; bins / hidden_layers without a compiled instance are
    refused by the library with a message that names what exists — nothing falls back silently."""
    with pytest.raises(NotImplementedError):
        mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=32, transforms=1,
                                    bins=8)
    gen = mf.generate.build_generator("nsf", input_features=2, output_features=2, hidden_layers=3, hidden_units=64,
                                      transforms=1, bins=12).to(backend)
    with pytest.raises(RuntimeError, match="no RQS kernel instance"):
        gen.sample_and_log_prob(8, z=torch.randn(8, 2).to(backend))


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
