"""The fused backward (parameter gradients inside the backward kernel, hand-scheduled MFMA chains — the default) against the two-kernel backward on identical inputs, on the GPU, at sizes where the
fused kernel is the one the library picks.  Both are fp32 with different summation orders:
agreement to 2e-5 of the largest gradient entry is required, ~5e-7 is typical.  The emulator runs of
test_flow_kernels.py cover the same kernel's indexing against the oracle; this test covers what the emulator cannot
see — the inline-asm scheduling (LDS prefetch distances, MFMA -> VALU read padding) on real hardware."""
import pytest
import torch

from conftest import set_bwd_variant

import mentflow_amd as mf


@pytest.mark.gpu
@pytest.mark.parametrize("d,bins,n", [(6, 20, 40000), (6, 20, 4097), (2, 20, 70000), (3, 8, 33000), (4, 20, 50001),
                                      (5, 8, 35000)])
def test_fused_backward_equals_two_kernel_backward(d, bins, n, monkeypatch):
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda", 0)
    res = []
    for fused in ("1", "0"):
        set_bwd_variant(monkeypatch, fused)
        torch.manual_seed(0)
        gen = mf.generate.build_generator("nsf", device=dev, input_features=d, output_features=d, hidden_layers=3,
                                          hidden_units=64, transforms=3, bins=bins)
        with torch.no_grad():
            for layer in gen.layers:
                lin = layer.linears()[-1]
                lin.weight.mul_(4.0)
                lin.bias.add_(torch.randn_like(lin.bias))
        torch.manual_seed(1)
        z = torch.randn(n, d, device=dev) * 1.5
        wx, wl = torch.randn(n, d, device=dev), torch.randn(n, device=dev)
        x, lp = gen.sample_and_log_prob(n, z=z)
        ((x * wx).sum() + (lp * wl).sum()).backward()
        res.append(torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).double())
    assert float((res[0] - res[1]).abs().max()) < 2e-5 * float(res[1].abs().max())
