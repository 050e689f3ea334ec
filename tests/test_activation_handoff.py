"""Activation hand-off from the training forward to the fused backward (mf_flow_rqs_layer_fwd_save /
mf_flow_rqs_layer_bwd_saved, include/mentflow_hip.h ABI 4).  The reference keeps every conditioner activation for autograd
(mentflow/generate/flows/zuko.py:24-26 -> zuko MaskedMLP); level 0 recomputes them in the backward, levels 1 / 2 load what the
forward stored.  Every level evaluates the same fp32 arithmetic on the same values (an fp32 MFMA chain is an exact, k-ordered
fma chain; the extra k-steps of a dense chain multiply masked-out zeros), so x and log_prob (one forward kernel) must agree BIT
FOR BIT and so must every gradient in the emulator build (compiled without floating-point contraction).  On the GPU the three
backward instances are separate compilations of the spline adjoint under hipcc's default -ffp-contract=fast, which fuses
multiply-adds across statements differently per instance: gradients there agree to fp32 rounding (measured 2e-7 of the largest
entry; gate 2e-6).  Through test_flow_kernels.py's oracle comparisons of the default level all levels are tied to the oracle."""
import pytest
import torch

import mentflow_amd as mf
from mentflow_amd import _lib


def _run(dev, d, bins, hidden_layers, n, level, seed=0):
    torch.manual_seed(seed)
    gen = mf.generate.build_generator("nsf", device=dev, input_features=d, output_features=d, hidden_layers=hidden_layers,
                                      hidden_units=64, transforms=3, bins=bins)
    with torch.no_grad():
        for layer in gen.layers:
            lin = layer.linears()[-1]
            lin.weight.mul_(4.0)
            lin.bias.add_(torch.randn_like(lin.bias))
    gen.spec().act_level = level
    torch.manual_seed(seed + 1)
    z = (torch.randn(n, d) * 1.5).to(dev).requires_grad_(True)
    wx, wl = torch.randn(n, d).to(dev), torch.randn(n).to(dev)
    x, lp = gen.sample_and_log_prob(n, z=z)
    ((x * wx).sum() + (lp * wl).sum()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()])
    return x.detach().cpu(), lp.detach().cpu(), g.cpu(), z.grad.cpu()


@pytest.mark.parametrize("d,bins,hidden_layers,n", [(6, 20, 3, 700), (6, 20, 3, 129), (2, 20, 3, 300), (3, 8, 2, 257),
                                                      (5, 20, 2, 333), (4, 8, 3, 96)])
def test_levels_agree_bitwise(backend, d, bins, hidden_layers, n):
    if backend.type == "cuda":
        n = n * 61 + 5                       # enough tiles for every workgroup column, ragged last tile and last group
    ref = _run(backend, d, bins, hidden_layers, n, 0)
    for level in (1, 2):
        out = _run(backend, d, bins, hidden_layers, n, level)
        for name, a, b in zip(("x", "log_prob", "parameter gradients", "dL/dz"), ref, out):
            err = float((a - b).abs().max())
            if backend.type == "cuda" and name in ("parameter gradients", "dL/dz"):
                assert err <= 2e-6 * float(a.abs().max()), f"level {level}: {name} off by {err:.3e} (largest entry {float(a.abs().max()):.3e})"
            else:
                assert torch.equal(a, b), f"level {level}: {name} differs from the recompute backward, max |diff| {err:.3e}"


def test_level_resolution_and_fallbacks(backend):
    """Run-time bins (no saved-activation instance) and the two-kernel backward fall back to level 0; the budget lowers the
    level; a level pinned above what fits is lowered, never an error."""
    lib = _lib.get_lib()
    gen = mf.generate.build_generator("nsf", device=backend, input_features=6, output_features=6, hidden_layers=3,
                                      hidden_units=64, transforms=2, bins=20)
    spec = gen.spec()
    assert spec.resolve_act_level(1000, backend) == 2
    per_layer2 = 4 * lib.mf_flow_rqs_act_floats(1000, 6, 3, 20, 2)
    per_layer1 = 4 * lib.mf_flow_rqs_act_floats(1000, 6, 3, 20, 1)
    # 32 tiles x 32 particles x (2 hidden levels x 64 floats [+ 5 features x 2 lane halves x 30 slots])
    assert per_layer1 == 32 * 32 * 2 * 64 * 4 and per_layer2 == per_layer1 + 32 * 32 * 5 * 60 * 4
    assert 4 * lib.mf_flow_rqs_act_floats(1000, 3, 2, 8, 2) == 32 * 32 * (64 + 2 * 24) * 4
    spec.act_budget_bytes = 2 * per_layer2 - 1
    assert spec.resolve_act_level(1000, backend) == 1
    spec.act_budget_bytes = 2 * per_layer1 - 1
    assert spec.resolve_act_level(1000, backend) == 0
    spec.act_budget_bytes = None
    spec.act_level = 1
    assert spec.resolve_act_level(1000, backend) == 1
    _lib.set_flow_bwd_variant(False)
    try:
        assert spec.resolve_act_level(1000, backend) == 0
    finally:
        _lib.set_flow_bwd_variant(None)
    gen13 = mf.generate.build_generator("nsf", device=backend, input_features=3, output_features=3, hidden_layers=3,
                                        hidden_units=64, transforms=1, bins=13)
    assert gen13.spec().resolve_act_level(1000, backend) == 0
    order = gen.spec().orders[0]
    assert lib.mf_flow_rqs_act_level(6, 3, 20, order) == 2 and lib.mf_flow_rqs_act_level(6, 3, 20, None) == 0


def test_save_entry_points_check_their_arguments(backend):
    lib = _lib.get_lib()
    gen = mf.generate.build_generator("nsf", device=backend, input_features=4, output_features=4, hidden_layers=3,
                                      hidden_units=64, transforms=1, bins=20)
    spec = gen.spec()
    from mentflow_amd import ops
    images = ops.pack_images(spec, gen.flat_parameters())
    n = 100
    x = torch.randn(n, 4, device=backend)
    y, lp = torch.empty_like(x), torch.empty(n, device=backend)
    need = lib.mf_flow_rqs_act_floats(n, 4, 3, 20, 2)
    small = torch.empty(need - 1, device=backend)
    with pytest.raises(RuntimeError, match="act buffer too small"):
        ops._layer_fwd(spec, 0, images[0], x, y, lp, lp, True, small, 2)
    with pytest.raises(RuntimeError, match="level must be 1 or 2"):
        _lib.call("mf_flow_rqs_layer_fwd_save", _lib.ptr(images[0]), 4, 3, 20, spec.orders[0], _lib.ptr(x), n, _lib.ptr(y),
                  _lib.ptr(lp), _lib.ptr(lp), 1, _lib.ptr(small), small.numel(), 3, _lib.stream_ptr(x))
