"""Size-independent properties at BASELINE.json's full per-GPU sizes (GPU only): the oracle cannot run there
(eager dense autograd needs ~29 GB at N = 100 k, P = 100), so the kernels are checked against invariants."""
import pytest
import torch

from conftest import set_bwd_variant

import mentflow_amd as mf
from mentflow_amd import ops
from mentflow_amd.harness import build_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    return torch.device("cuda", 0)


def test_c4_histograms_additive_normalised_and_chunk_invariant(dev):
    """C4 shapes: d = 6, P = 100, B = 64, 2 097 152 particles."""
    prob = build_problem(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", meas_samples=200_000, penalty_parameter=500.0)
    diag = prob.diagnostics[0][0]
    V = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
    n = 2_097_152
    torch.manual_seed(0)
    x = torch.randn(n, 6, device=dev) * 1.2
    R = ops.kde_radius(0.5)
    S = ops.ProjKde1dFn.apply(x, V, diag.coords, float(diag.bandwidth), R)
    Sa = ops.ProjKde1dFn.apply(x[: n // 3].contiguous(), V, diag.coords, float(diag.bandwidth), R)
    Sb = ops.ProjKde1dFn.apply(x[n // 3:].contiguous(), V, diag.coords, float(diag.bandwidth), R)
    # linearity in the particle set (fixed-point accumulation: exact up to the final fp32 rounding)
    torch.testing.assert_close(S, Sa + Sb, rtol=3e-7, atol=1e-6)
    # run-to-run reproducibility of the integer accumulation
    S2 = ops.ProjKde1dFn.apply(x, V, diag.coords, float(diag.bandwidth), R)
    assert (S - S2).abs().max() <= 1e-6 * S.abs().max()
    # mass: every particle inside the range contributes sum_k exp(-2 (t-k)^2) ~ sqrt(pi/2) = 1.2533 (sigma = delta/2)
    u = x @ V.T
    inside = ((u > -3.0) & (u < 3.0)).float().sum(0)
    assert (S.sum(1) >= 1.2533 * inside * 0.999).all() and (S.sum(1) <= 1.2534 * n).all()
    # normalised prediction integrates to one; histogram of a projection equals the single-projection call
    ghat, _ = ops.HistNormDiscFn.apply(S, None, True, 1.0 / n, float(diag.resolution), 1e-10, 0, 0.0, 1.0)
    torch.testing.assert_close((ghat.sum(1) * diag.resolution).cpu(), torch.ones(100), rtol=1e-5, atol=1e-5)
    one = mf.simulate.forward(x, [prob.transforms[7]], [[diag]])[0][0]
    torch.testing.assert_close(one, ghat[7], rtol=1e-6, atol=1e-9)


def test_c3_flow_roundtrip_logprob_and_chunked_backward(dev, monkeypatch):
    """C3/C4 flow: 6-D NSF 5 x [3 x 64], K = 20; 1 048 576 particles."""
    torch.manual_seed(0)
    gen = mf.generate.build_generator("nsf", device=dev, input_features=6, output_features=6, hidden_layers=3,
                                      hidden_units=64, transforms=5, bins=20)
    with torch.no_grad():                       # make the conditioner matter (default init is near identity)
        for layer in gen.layers:
            lin = layer.linears()[-1]
            lin.weight.mul_(2.0)
    n = 1_048_576
    z = torch.randn(n, 6, device=dev)
    with torch.no_grad():
        x, lp = gen.sample_and_log_prob(n, z=z)
    assert torch.isfinite(x).all() and torch.isfinite(lp).all()
    # encode -> decode round trip, and log_prob(x) == log_prob returned with the sample
    zb = gen.inverse(x)
    err = (zb - z).abs()
    assert err.median() < 2e-6 and err.quantile(0.999) < 1e-3
    lp2 = gen.log_prob(x)
    dl = (lp2 - lp).abs()
    assert dl.median() < 2e-5 and dl.quantile(0.999) < 5e-3
    # log-density normalisation proxy: E_z[exp(-ladj)] is finite and ladj has the right sign convention:
    # logp = logN(z) - ladj  =>  ladj = logN(z) - logp
    ladj = (-0.5 * (z ** 2).sum(1) - 3 * 1.8378770664093453) - lp
    assert torch.isfinite(ladj).all()
    # backward: neither the variant (fused kernel, one launch per layer / two kernels with a scratch hand-off) nor the
    # chunk size of the two-kernel path (scratch reuse, tile padding, atomics) may change the gradients
    w = torch.randn(n, 6, device=dev)
    grads = []
    for fused, chunk in (("1", 1 << 19), ("0", 1 << 19), ("0", 100_003)):
        set_bwd_variant(monkeypatch, fused)
        gen.spec().bwd_chunk = chunk
        gen.zero_grad()
        xx, ll = gen.sample_and_log_prob(n, z=z)
        ((xx * w).sum() / n + ll.mean()).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).clone())
    set_bwd_variant(monkeypatch, None)
    gen.spec().bwd_chunk = 1 << 19
    for other in grads[1:]:
        torch.testing.assert_close(grads[0], other, rtol=2e-4, atol=2e-5 * float(grads[0].abs().max()))
    # linearity of the backward in the upstream gradient
    gen.zero_grad()
    xx, ll = gen.sample_and_log_prob(n, z=z)
    (2.0 * ((xx * w).sum() / n + ll.mean())).backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in gen.parameters()])
    torch.testing.assert_close(g2, 2.0 * grads[0], rtol=2e-4, atol=4e-5 * float(grads[0].abs().max()))


def test_c5_2d_histograms_additive(dev):
    """C5 shapes: 100 two-dimensional projections, 85 x 85 bins."""
    from mentflow_amd.harness import make_transforms_nd_2d_random
    tfs = [t.to(dev) for t in make_transforms_nd_2d_random(100, 6, 0)]
    e = torch.linspace(-3.5, 3.5, 86)
    diag = mf.diagnostics.Histogram2D(axis=(0, 2), edges=(e, e), bandwidth=(0.5, 0.5)).to(dev)
    V0 = torch.stack([t.matrix[0] for t in tfs]).contiguous()
    V1 = torch.stack([t.matrix[2] for t in tfs]).contiguous()
    n = 262_144
    torch.manual_seed(1)
    x = torch.randn(n, 6, device=dev)
    args = (diag.coords_x, diag.coords_y, float(diag.bandwidth_x), float(diag.bandwidth_y), 4, 4)
    S = ops.ProjKde2dFn.apply(x, V0, V1, *args)
    Sa = ops.ProjKde2dFn.apply(x[: n // 2].contiguous(), V0, V1, *args)
    Sb = ops.ProjKde2dFn.apply(x[n // 2:].contiguous(), V0, V1, *args)
    torch.testing.assert_close(S, Sa + Sb, rtol=3e-7, atol=1e-6)
    # marginalising the 2-D kernel sums over one axis gives (1-D sums) x (mass of the other kernel ~ 1.2533)
    S1 = ops.ProjKde1dFn.apply(x, V0, diag.coords_x, float(diag.bandwidth_x), 4)
    u1 = x @ V1.T
    inside = (u1.abs() < 3.0).all(0)
    ratio = S.sum(2)[inside] / S1[inside].clamp_min(1e-3)
    big = S1[inside] > 100.0
    assert ((ratio[big] - 1.2533).abs() < 5e-3).all()


def test_c2_full_size_flow_and_histograms(dev, monkeypatch):
    """C2 at its full size: d = 2 NSF 5 x [3 x 64] K = 20, 7 rotations x 85 bins, 1 048 576 particles."""
    prob = build_problem(ndim=2, num=7, bins=85, xmax=3.5, seed=21, transforms=5, prior_scale=1.0, device=dev,
                         dist_name="swissroll", optics="2d_linear", gen_name="nsf", meas_samples=200_000,
                         penalty_parameter=500.0)
    gen = prob.model.generator
    n = 1_048_576
    torch.manual_seed(5)
    z = torch.randn(n, 2, device=dev)
    with torch.no_grad():
        x, lp = gen.sample_and_log_prob(n, z=z)
    assert torch.isfinite(x).all() and torch.isfinite(lp).all()
    # round trip and density consistency
    err = (gen.inverse(x) - z).abs()
    assert err.median() < 2e-6 and err.quantile(0.999) < 1e-3
    dl = (gen.log_prob(x) - lp).abs()
    assert dl.median() < 2e-5 and dl.quantile(0.999) < 5e-3
    # histograms: additive over a split of the batch, normalised to one
    diag = prob.diagnostics[0][0]
    V = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
    R = ops.kde_radius(0.5)
    S = ops.ProjKde1dFn.apply(x, V, diag.coords, float(diag.bandwidth), R)
    Sa = ops.ProjKde1dFn.apply(x[:400_001].contiguous(), V, diag.coords, float(diag.bandwidth), R)
    Sb = ops.ProjKde1dFn.apply(x[400_001:].contiguous(), V, diag.coords, float(diag.bandwidth), R)
    torch.testing.assert_close(S, Sa + Sb, rtol=3e-7, atol=1e-6)
    # the whole loss: fused vs two-kernel backward, and bitwise run-to-run reproducibility of the parameter gradients
    grads = []
    for fused in ("1", "1", "0"):
        set_bwd_variant(monkeypatch, fused)
        gen.inject_z = z
        prob.model.zero_grad()
        L, H, D = prob.model.loss(n)
        L.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).clone())
    set_bwd_variant(monkeypatch, None)
    assert torch.equal(grads[0], grads[1]), "parameter gradients are not bitwise reproducible"
    torch.testing.assert_close(grads[0], grads[2], rtol=2e-4, atol=2e-5 * float(grads[0].abs().max()))


def test_16m_particles_indexing_smoke(dev):
    """C4's GLOBAL batch on one GPU (the strong-scaling N = 1 point): 16 777 216 particles through the flow forward,
    the 100-projection KDE forward / backward and the fused flow backward.  Checks the 64-bit indexing and the grid-stride
    arithmetic against the same computation done in 8 shards of 2 097 152 (what 8 ranks would each do)."""
    prob = build_problem(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", meas_samples=200_000, penalty_parameter=500.0)
    gen = prob.model.generator
    diag = prob.diagnostics[0][0]
    V = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
    n, shards = 16_777_216, 8
    m = n // shards
    torch.manual_seed(9)
    z = torch.randn(n, 6, device=dev)
    R = ops.kde_radius(0.5)
    coef = torch.randn(100, 64, device=dev)

    def run(zz):
        gen.zero_grad()
        zz = zz.contiguous()
        x, lp = gen.sample_and_log_prob(zz.shape[0], z=zz)
        xd = x.detach().requires_grad_(True)
        S = ops.ProjKde1dFn.apply(xd, V, diag.coords, float(diag.bandwidth), R)
        (S * coef).sum().backward()
        gx = xd.grad
        ((x * gx).sum() / n + lp.sum() / n).backward()
        g = torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).double()
        return x.detach(), lp.detach(), S.detach().double(), gx, g

    x, lp, S, gx, g = run(z)
    assert torch.isfinite(x).all() and torch.isfinite(lp).all() and torch.isfinite(gx).all()
    S_sum = torch.zeros_like(S)
    g_sum = torch.zeros_like(g)
    for k in range(shards):
        xs, lps, Ss, gxs, gs = run(z[k * m:(k + 1) * m])
        assert torch.equal(xs, x[k * m:(k + 1) * m]) and torch.equal(lps, lp[k * m:(k + 1) * m])
        assert torch.equal(gxs, gx[k * m:(k + 1) * m])
        S_sum += Ss
        g_sum += gs
    torch.testing.assert_close(S, S_sum, rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(g, g_sum, rtol=1e-4, atol=1e-5 * float(g.abs().max()))
    # the tail end of the batch really was processed (last particle's row is not an uninitialised buffer)
    assert float(gx[-1].abs().sum()) > 0 and float(x[-1].abs().sum()) > 0


def test_c5_full_size_step_is_reproducible_and_variant_independent(dev, monkeypatch):
    """C5 at its full per-GPU size: 100 two-dimensional projections x 85 x 85 bins, 2 097 152 particles, the whole
    MENTFlow.loss() + backward: bitwise reproducible, histograms additive over a split of the batch, and parameter
    gradients independent of the flow-backward variant."""
    prob = build_problem(ndim=6, num=100, bins=85, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", optics="nd_2d_random", meas_samples=200_000, penalty_parameter=500.0)
    gen = prob.model.generator
    n = 2_097_152
    torch.manual_seed(3)
    z = torch.randn(n, 6, device=dev)
    runs = []
    for fused in ("1", "1", "0"):
        set_bwd_variant(monkeypatch, fused)
        gen.inject_z = z
        prob.model.zero_grad()
        L, H, D = prob.model.loss(n)
        L.backward()
        runs.append((L.detach().clone(), H.detach().clone(), torch.stack(D).detach().clone(),
                     torch.cat([p.grad.reshape(-1) for p in gen.parameters()]).clone()))
    set_bwd_variant(monkeypatch, None)
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), "C5 step is not bitwise reproducible"
    assert torch.equal(runs[0][0], runs[2][0]) and torch.equal(runs[0][2], runs[2][2])      # forward is the same code
    torch.testing.assert_close(runs[0][3], runs[2][3], rtol=2e-4, atol=2e-5 * float(runs[0][3].abs().max()))
    assert torch.isfinite(runs[0][3]).all() and float(runs[0][3].abs().max()) > 0
    # 2-D histograms of the full batch = sum over two shards
    with torch.no_grad():
        x, _ = gen.sample_and_log_prob(n, z=z)
    diag = prob.diagnostics[0][0]
    V0 = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
    V1 = torch.stack([t.matrix[2] for t in prob.transforms]).contiguous()
    args = (diag.coords_x, diag.coords_y, float(diag.bandwidth_x), float(diag.bandwidth_y), 4, 4)
    S = ops.ProjKde2dFn.apply(x, V0, V1, *args)
    Sa = ops.ProjKde2dFn.apply(x[:700_001].contiguous(), V0, V1, *args)
    Sb = ops.ProjKde2dFn.apply(x[700_001:].contiguous(), V0, V1, *args)
    torch.testing.assert_close(S, Sa + Sb, rtol=3e-7, atol=1e-6)


def test_kde_results_do_not_depend_on_launch_shape():
    """The tuning knobs (block size, LDS budget, particles per workgroup; read once per process) must not change a single
    bit of the histograms (integer accumulation).  The backward sums a particle's projections in registers: its result may
    only move by fp32 reassociation when the number of lanes sharing a particle changes with the block size."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "kde_sweep.py")
    outs = {}
    for kind, cfgs in (("1d", [{}, {"MENTFLOW_KDE1D_BLOCK": "256", "MENTFLOW_KDE1D_WAVES": "8", "MENTFLOW_KDE1D_BWD_BLOCK": "1024"},
                               {"MENTFLOW_KDE1D_BLOCK": "512", "MENTFLOW_KDE1D_LDS": "159000", "MENTFLOW_KDE1D_WAVES": "2"}]),
                       ("2d", [{}, {"MENTFLOW_KDE2D_BLOCK": "256", "MENTFLOW_KDE2D_FWD_LDS": "118000", "MENTFLOW_KDE2D_BWD_BLOCK": "256",
                                    "MENTFLOW_KDE2D_BWD_NPT": "2"}])):
        for cfg in cfgs:
            env = dict(os.environ, **cfg)
            r = subprocess.run([sys.executable, tool, "--kind", kind, "--n", "300000"], env=env, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            j = json.loads(r.stdout.strip().splitlines()[-1])
            outs.setdefault(kind, []).append((j["sumS"], j["sum|gx|"]))
    for kind, vals in outs.items():
        assert len({v[0] for v in vals}) == 1, (kind, vals)
        for v in vals[1:]:
            assert abs(v[1] - vals[0][1]) <= 1e-6 * abs(vals[0][1]), (kind, vals)
