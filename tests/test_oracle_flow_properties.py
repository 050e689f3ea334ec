"""Property tests of oracle/flow.py (zuko 1.3.1 restatement — parity unpinned, see oracle/__init__.py).
No golden vectors exist for the flow; SURVEY.md §8c-2 lists the properties that stand in for them."""
import math

import pytest
import torch

from oracle import flow as of


@pytest.mark.parametrize("d,kind", [(2, "rqs"), (6, "rqs"), (2, "affine"), (6, "affine")])
def test_masks_autoregressive_structure(d, kind):
    spec = of.init_flow(d, (64, 64, 64), transforms=2, kind=kind, bins=20, seed=0)
    for layer in spec.layers:
        # effective dependency of output i on input j through the product of masks
        reach = layer.masks[0].float()
        for m in layer.masks[1:]:
            reach = (m.float() @ reach).clamp(max=1)
        reach = reach.view(d, spec.total, d)
        for i in range(d):
            for j in range(d):
                expect = 1.0 if layer.order[i] > layer.order[j] else 0.0
                assert (reach[i, :, j] == expect).all(), (i, j)
        # hidden units cycle over the d-1 non-empty dependency classes
        assert layer.masks[0].shape == (64, d) and layer.masks[-1].shape == (d * spec.total, 64)


def test_param_count_matches_survey():
    spec = of.init_flow(6, (64, 64, 64), transforms=5, kind="rqs", bins=20, seed=0)
    assert sum(p.numel() for p in spec.parameters()) == 158890
    spec = of.init_flow(2, (64, 64, 64), transforms=5, kind="rqs", bins=20, seed=0)
    assert sum(p.numel() for p in spec.parameters()) == 80910


@pytest.mark.parametrize("d,kind", [(2, "rqs"), (6, "rqs"), (6, "affine")])
def test_invertibility_and_ladj_vs_autograd_jacobian(d, kind):
    spec = of.init_flow(d, (64, 64, 64), transforms=3, kind=kind, bins=20, seed=1, dtype=torch.float64)
    # make the conditioner non-trivial (default init gives near-identity splines)
    for layer in spec.layers:
        layer.weights[-1].mul_(4.0)
        layer.biases[-1].add_(torch.randn_like(layer.biases[-1]))
    torch.manual_seed(3)
    z = torch.randn(16, d, dtype=torch.float64) * 1.5
    x, ladj = of.flow_forward(z, spec)
    zb = of.flow_inverse(x, spec)
    assert (zb - z).abs().max() < 1e-8
    for n in range(4):
        J = torch.autograd.functional.jacobian(lambda v: of.flow_forward(v[None], spec)[0][0], z[n])
        assert abs(torch.linalg.slogdet(J)[1] - ladj[n]) < 1e-9
    # log_prob(x) == logN(z) - ladj
    lp = of.log_prob(x, spec)
    _, lp2 = of.sample_and_log_prob(z, spec)
    assert (lp - lp2).abs().max() < 1e-8


def test_rqs_identity_outside_bound_and_monotone_knots():
    torch.manual_seed(0)
    phi = torch.randn(64, 59, dtype=torch.float64) * 3
    X, Y, D = of.rqs_knots(phi, 20)
    assert torch.allclose(X[:, 0], torch.full((64,), -5.0, dtype=torch.float64))
    assert torch.allclose(X[:, -1], torch.full((64,), 5.0, dtype=torch.float64), atol=1e-12)
    assert (X[:, 1:] > X[:, :-1]).all() and (Y[:, 1:] > Y[:, :-1]).all()
    assert (D > 1e-3).all() and (D < 1e3).all() and (D[:, 0] == 1).all() and (D[:, -1] == 1).all()
    x = torch.tensor([-7.0, 5.5, 100.0], dtype=torch.float64)
    y, l = of.rqs_forward(x, phi[:3], 20)
    assert torch.equal(y, x) and (l == 0).all()
    assert torch.equal(of.rqs_inverse(x, phi[:3], 20), x)


def test_rqs_gradcheck_fp64():
    torch.manual_seed(0)
    phi = (torch.randn(5, 59, dtype=torch.float64)).requires_grad_(True)
    x = (torch.randn(5, dtype=torch.float64) * 2).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b: of.rqs_forward(a, b, 20), (x, phi), eps=1e-6, atol=1e-6)


def test_affine_restated_formula():
    phi = torch.tensor([[0.3, 2.0], [-1.0, -50.0]], dtype=torch.float64)
    x = torch.tensor([1.5, -0.2], dtype=torch.float64)
    y, l = of.affine_forward(x, phi)
    ls = phi[:, 1] / (1 + (phi[:, 1] / math.log(1e-3)).abs())
    assert torch.allclose(y, x * ls.exp() + phi[:, 0]) and torch.allclose(l, ls)
    assert torch.allclose(of.affine_inverse(y, phi), x)
    assert (ls.abs() < abs(math.log(1e-3))).all()
