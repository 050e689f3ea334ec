"""Autograd contract of AutoregressiveFlow (see its docstring): gradient with respect to the base draw z against the oracle
(reference: mentflow/generate/flows/zuko.py:28-29, zuko's transform is differentiable in z), the default deposit-into-.grad
behaviour, the autograd_parameters switch, and the flat-storage validity check."""
import pytest
import torch

import mentflow_amd as mf
from oracle import flow as of
from oracle.harness import flow_spec_from_generator


def _gen(dev, kind="nsf", d=6, transforms=3):
    torch.manual_seed(4)
    return mf.generate.build_generator(kind, device=dev, input_features=d, output_features=d, hidden_layers=3, hidden_units=64,
                                       transforms=transforms, **({"bins": 20} if kind == "nsf" else {}))


def _oracle(gen, z, wx, wl):
    spec = flow_spec_from_generator(gen, torch.float64)
    params = spec.parameters()
    for p in params:
        p.requires_grad_(True)
    zo = z.detach().cpu().double().requires_grad_(True)
    x, lp = of.sample_and_log_prob(zo, spec)
    ((x * wx.cpu().double()).sum() + (lp * wl.cpu().double()).sum()).backward()
    return zo.grad, torch.cat([p.grad.reshape(-1) for p in params])


@pytest.mark.parametrize("kind,d", [("nsf", 6), ("nsf", 2), ("maf", 2)])
def test_gradient_with_respect_to_the_base_draw(backend, kind, d):
    gen = _gen(backend, kind, d)
    torch.manual_seed(8)
    z = torch.randn(1500, d, device=backend, requires_grad=True)
    wx, wl = torch.randn(1500, d, device=backend), torch.randn(1500, device=backend)
    x, lp = gen.sample_and_log_prob(1500, z=z)
    ((x * wx).sum() + (lp * wl).sum()).backward()
    gz_o, gp_o = _oracle(gen, z, wx, wl)
    assert z.grad is not None
    assert float((z.grad.cpu().double() - gz_o).abs().max() / gz_o.abs().max()) < 2e-4
    gp = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).cpu().double()
    assert float((gp - gp_o).abs().max() / gp_o.abs().max()) < 5e-4
    # forward(z) is the same transform (flows/zuko.py:28-29) and is differentiable in z too, parameters frozen or not
    z2 = z.detach().clone().requires_grad_(True)
    for p in gen.parameters():
        p.requires_grad_(False)
    (gen.forward(z2) * wx).sum().backward()
    for p in gen.parameters():
        p.requires_grad_(True)
    assert z2.grad is not None and torch.isfinite(z2.grad).all()


def test_parameters_are_not_autograd_inputs_by_default_and_the_switch_makes_them(backend):
    gen = _gen(backend)
    z = torch.randn(800, 6, device=backend)
    x, lp = gen.sample_and_log_prob(800, z=z)
    loss = x.square().mean() + lp.mean()
    with pytest.raises(RuntimeError, match="not have been used in the graph|appears to not have been used"):
        torch.autograd.grad(loss, list(gen.parameters()), retain_graph=True)
    loss.backward()                                                  # ... but backward() deposits every .grad
    g_default = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).clone()
    assert g_default.abs().max() > 0
    gen.zero_grad()
    gen.autograd_parameters = True
    fired = []
    first = next(gen.parameters())
    first.register_hook(lambda g: fired.append(g.shape))             # per-parameter hooks fire in this mode
    x, lp = gen.sample_and_log_prob(800, z=z)
    loss = x.square().mean() + lp.mean()
    grads = torch.autograd.grad(loss, list(gen.parameters()))
    g_switch = torch.cat([g.reshape(-1) for g in grads])
    assert fired and all(p.grad is None for p in gen.parameters())   # autograd.grad does not touch .grad
    torch.testing.assert_close(g_switch, g_default, rtol=0, atol=0)  # same kernels, same values


def test_flat_storage_check_sees_a_replaced_middle_parameter(backend):
    gen = _gen(backend)
    z = torch.randn(600, 6, device=backend)
    x0, _ = gen.sample_and_log_prob(600, z=z)
    ps = list(gen.parameters())
    assert gen._flat_ok()
    with torch.no_grad():
        ps[3].data = ps[3].data.clone() * 1.5                        # storage of a MIDDLE parameter replaced
    assert not gen._flat_ok()
    x1, _ = gen.sample_and_log_prob(600, z=z)                        # re-flattened: the kernels see the new values
    spec = flow_spec_from_generator(gen, torch.float64)
    xo, _ = of.sample_and_log_prob(z.cpu().double(), spec)
    assert (x1.detach().cpu().double() - xo).abs().max() < 1e-4
    assert (x1 - x0).abs().max() > 1e-4
    # a swapped Parameter object is seen too
    lin = gen.layers[1].linears()[1]
    lin.bias = torch.nn.Parameter(torch.zeros_like(lin.bias))
    assert not gen._flat_ok()
    gen.sample_and_log_prob(600, z=z)
    assert gen._flat_ok()
    # non-float32 parameters are refused, not silently cast
    gen.double()
    with pytest.raises(TypeError, match="float32"):
        gen.sample_and_log_prob(600, z=z)
    gen.float()
    gen.sample_and_log_prob(600, z=z)
