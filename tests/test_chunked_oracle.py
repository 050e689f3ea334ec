"""The N-chunked two-pass fp64 oracle (oracle/chunked.py) against the dense fp64 oracle (oracle.harness.oracle_step) at
sizes where both run (CPU; the problems are built on the emulated kernels), and — with -m gpu — the HIP path against the
chunked oracle at BASELINE.json's sizes: an INDEPENDENT value for the 2 M / 4 M-particle accumulations (slab rows -> fp64
reduce, the fixed-point histogram flush at shift 8, per-workgroup entropy partials) that the dense oracle cannot reach."""
import time

import pytest
import torch

from mentflow_amd.harness import build_problem
from oracle.chunked import chunked_step
from oracle.harness import oracle_step


@pytest.fixture
def emu(emu_library):
    from mentflow_amd import _lib
    _lib.use_library(emu_library)
    return torch.device("cpu")


@pytest.mark.parametrize("optics,num,bins,ndim", [("nd_1d", 5, 32, 6), ("nd_2d_random", 3, 21, 6), ("2d_linear", 4, 40, 2)])
def test_chunked_equals_dense_oracle(emu, optics, num, bins, ndim):
    prob = build_problem(ndim=ndim, num=num, bins=bins, xmax=3.5, seed=3, transforms=2, prior_scale=1.5, device=emu,
                         dist_name="gaussian_mixture" if ndim == 6 else "swissroll", optics=optics, meas_samples=20000,
                         penalty_parameter=250.0)
    torch.manual_seed(5)
    z = torch.randn(2500, ndim)
    Lo, Ho, Do, go = oracle_step(prob, z, torch.float64)
    for chunk in (700, 4096):                       # ragged last chunk / a single chunk
        r = chunked_step(prob, z, chunk=chunk)
        assert abs(float(r.L) - float(Lo)) < 1e-11 * max(1.0, abs(float(Lo)))
        assert abs(float(r.H) - float(Ho)) < 1e-12 * max(1.0, abs(float(Ho)))
        assert (torch.stack(r.D) - torch.stack(Do)).abs().max() < 1e-13
        assert (r.grad - go).abs().max() < 1e-10 * float(go.abs().max())
    r = chunked_step(prob, z, chunk=1000, backward=False)
    assert r.grad is None and abs(float(r.L) - float(Lo)) < 1e-11 * max(1.0, abs(float(Lo)))


# ----------------------------------------------------------------------------------------------------------- GPU, full size
def _gpu():
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    return torch.device("cuda", 0)


def _progress(msg):
    """A line per 16 chunks on the real stdout: a GPU-box run that stays silent for minutes is taken to be hung."""
    import sys
    print("  " + msg, file=sys.__stdout__, flush=True)


def _forward_vs_chunked(prob, n, dev, chunk, label):
    """Forward quantities of one loss() at n particles against the chunked fp64 oracle.  Gates = the small-batch ones
    (tests/test_baseline_configs.py): H 2e-5 rel, D 2e-4 rel, L 1e-4 + mu 2e-6, histograms rtol 2e-5 + atol 1e-6."""
    import mentflow_amd as mf
    d = prob.cfg["ndim"]
    g = torch.Generator().manual_seed(77)
    z = torch.randn(n, d, generator=g)
    prob.model.generator.inject_z = z.to(dev)
    t0 = time.time()
    with torch.no_grad():
        L, H, D = prob.model.loss(n)
        x = prob.model.generator.forward(z.to(dev))
        preds = [row[0] for row in mf.simulate.forward(x, prob.transforms, prob.diagnostics)]
    torch.cuda.synchronize()
    t1 = time.time()
    r = chunked_step(prob, z, chunk=chunk, backward=False, progress=_progress)
    t2 = time.time()
    mu = float(prob.model.penalty_parameter)
    Dk, Dr = torch.stack(D).cpu().double(), torch.stack(r.D)
    eH = abs(float(H) - float(r.H))
    eD = float((Dk - Dr).abs().max())
    eL = abs(float(L) - float(r.L))
    eP = max(float(((p.cpu().double() - q).abs() - 2e-5 * q.abs()).max()) for p, q in zip(preds, r.predictions))
    print(f"\n[{label}] n={n}: gpu {t1 - t0:.1f} s, chunked fp64 oracle {t2 - t1:.1f} s; |dH|={eH:.2e} max|dD|={eD:.2e} "
          f"|dL|={eL:.2e} (L={float(r.L):.6f}) hist excess over rtol 2e-5: {eP:.2e}")
    assert eH < 2e-5 * max(1.0, abs(float(r.H)))
    assert eD < 2e-6 + 2e-4 * float(Dr.abs().max())
    assert eL < 1e-4 + mu * 2e-6 + 1e-5 * abs(float(r.L))
    assert eP < 1e-6
    return r


@pytest.mark.gpu
def test_c4_full_size_forward_vs_chunked_fp64_oracle():
    """C4's per-GPU shard: 2 097 152 particles, 100 projections x 64 bins — S, H, D, L against fp64."""
    dev = _gpu()
    prob = build_problem(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", meas_samples=1_000_000, penalty_parameter=500.0)
    _forward_vs_chunked(prob, 2_097_152, dev, chunk=16384, label="C4")


@pytest.mark.gpu
def test_c3_full_size_forward_vs_chunked_fp64_oracle():
    """C3 at its full 4 194 304 particles (25 projections x 64 bins, rings)."""
    dev = _gpu()
    prob = build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=dev,
                         dist_name="rings", meas_samples=1_000_000, penalty_parameter=500.0)
    _forward_vs_chunked(prob, 4_194_304, dev, chunk=32768, label="C3")


def _grads_vs_chunked(prob, n, dev, chunk, label, gtol=5e-4):
    d = prob.cfg["ndim"]
    g = torch.Generator().manual_seed(78)
    z = torch.randn(n, d, generator=g)
    prob.model.generator.inject_z = z.to(dev)
    prob.model.zero_grad()
    L, H, D = prob.model.loss(n)
    L.backward()
    gk = torch.cat([p.grad.reshape(-1) for p in prob.model.parameters()]).cpu().double()
    t0 = time.time()
    r = chunked_step(prob, z, chunk=chunk, backward=True, progress=_progress)
    mu = float(prob.model.penalty_parameter)
    eg = float((gk - r.grad).abs().max() / r.grad.abs().max())
    eL = abs(float(L.detach()) - float(r.L))
    print(f"\n[{label}] n={n}: chunked fp64 oracle fwd+bwd {time.time() - t0:.1f} s; |dL|={eL:.2e} "
          f"parameter-gradient error {eg:.2e} of the largest entry")
    assert eL < 1e-4 + mu * 2e-6 + 1e-5 * abs(float(r.L))
    assert eg < gtol, f"parameter-gradient error {eg:.2e} of the largest entry (gate {gtol:.0e})"


@pytest.mark.gpu
def test_c4_gradients_262144_vs_chunked_fp64_oracle():
    dev = _gpu()
    prob = build_problem(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", meas_samples=1_000_000, penalty_parameter=500.0)
    _grads_vs_chunked(prob, 262_144, dev, chunk=8192, label="C4 grads")


@pytest.mark.gpu
def test_c5_gradients_131072_vs_chunked_fp64_oracle():
    """C5: 100 2-D projections (85 x 85 bins) at 131 072 particles, loss and parameter gradients."""
    dev = _gpu()
    prob = build_problem(ndim=6, num=100, bins=85, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", optics="nd_2d_random", meas_samples=1_000_000,
                         penalty_parameter=500.0)
    _grads_vs_chunked(prob, 131_072, dev, chunk=8192, label="C5 grads")
