"""ORACLE (test infrastructure) — generates tests/golden/ref_*.npz by running the REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference; never runs on the GPU box):

    python -m oracle.gen_golden

The reference package cannot be imported whole (mentflow/__init__.py pulls zuko, POT, scikit-image,
psdist, ultraplot — none installed, no network), so its hot-path modules are imported by path behind an
empty package shell with inert stand-in modules for names that are only touched at import time
(SURVEY.md §8c / Appendix C.1).  None of the stand-ins is ever *called* on the paths exercised here.
Only inputs and expected outputs (data) are written; no reference source travels.

What cannot be generated: anything through zuko (the flow itself) — see oracle/flow.py ("parity unpinned").
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF = "/root/reference"


def _boot():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("mentflow")
    pkg.__path__ = [os.path.join(REF, "mentflow")]
    sys.modules["mentflow"] = pkg
    sys.modules["ot"] = types.ModuleType("ot")
    zk = types.ModuleType("zuko")
    zk.flows = types.ModuleType("zuko.flows")
    zk.flows.Flow = object
    sys.modules["zuko"] = zk
    sys.modules["zuko.flows"] = zk.flows
    sys.modules["skimage"] = types.ModuleType("skimage")
    import mentflow.core, mentflow.simulate, mentflow.diagnostics, mentflow.entropy  # noqa
    import mentflow.loss, mentflow.prior, mentflow.generate, mentflow.distributions  # noqa
    import mentflow
    return mentflow


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def main():
    mf = _boot()
    from mentflow.diagnostics.histogram import kde_histogram_1d, kde_histogram_2d
    from mentflow.diagnostics import Histogram1D, Histogram2D
    from mentflow.simulate import LinearTransform, forward
    from mentflow.entropy import MonteCarloEntropyEstimator
    from mentflow.prior import Gaussian
    from mentflow.loss import kl_divergence, mean_absolute_error, mean_square_error
    from mentflow.core import MENTFlow
    from mentflow.generate import GenerativeModel

    g = torch.Generator().manual_seed(20250620)
    randn = lambda *s: torch.randn(*s, generator=g)

    # ---- 1. LinearTransform
    x = randn(512, 6)
    M = torch.eye(6)
    M[0, :] = torch.nn.functional.normalize(randn(6), dim=0)
    M[3, :] = randn(6)
    t = LinearTransform(M)
    u = t(x)
    _save("ref_linear_transform", x=_np(x), M=_np(M), u=_np(u), x_back=_np(t.inverse(u)))

    # ---- 2. kde 1d (values + grad), incl. out-of-range particles
    for bins, xmax in ((64, 4.0), (85, 3.5)):
        edges = torch.linspace(-xmax, xmax, bins + 1)
        res = edges[1] - edges[0]
        bw = 0.5 * res
        uu = (randn(4096) * 1.7).requires_grad_(True)
        w = randn(bins)
        hist = kde_histogram_1d(uu, edges, bandwidth=bw)
        (hist * w).sum().backward()
        _save(f"ref_kde1d_B{bins}", u=_np(uu), edges=_np(edges), bandwidth=_np(bw), w=_np(w),
              hist=_np(hist), grad_u=_np(uu.grad))

    # ---- 3. kde 2d
    for bins, xmax in ((64, 4.0), (85, 3.5)):
        ex = torch.linspace(-xmax, xmax, bins + 1)
        ey = torch.linspace(-xmax * 0.9, xmax * 0.9, bins + 1)
        bwx, bwy = 0.5 * (ex[1] - ex[0]), 0.5 * (ey[1] - ey[0])
        uu = (randn(2048, 2) * 1.5).requires_grad_(True)
        w = randn(bins, bins)
        hist = kde_histogram_2d(uu[:, 0], uu[:, 1], bins=(ex, ey), bandwidth=(bwx, bwy))
        (hist * w).sum().backward()
        _save(f"ref_kde2d_B{bins}", u=_np(uu), edges_x=_np(ex), edges_y=_np(ey), bandwidth_x=_np(bwx),
              bandwidth_y=_np(bwy), w=_np(w), hist=_np(hist), grad_u=_np(uu.grad))

    # ---- 4. hard-binned measurement generation + renormalisation (experiments/setup.py:52-73)
    xs = randn(6000, 6) * 1.3
    xs[0, 0] = 4.0            # exactly on the last edge (right-inclusive)
    xs[1, 0] = -4.0           # exactly on the first edge
    xs[2, 0] = 7.0            # out of range
    edges = torch.linspace(-4.0, 4.0, 65)
    d1 = Histogram1D(axis=0, edges=edges, bandwidth=0.5, noise=True, noise_scale=0.0)
    d1.kde = False
    h1 = d1(xs)
    m1 = h1 / torch.sum(h1) / (d1.edges[1] - d1.edges[0])
    e2 = [torch.linspace(-3.5, 3.5, 33), torch.linspace(-3.5, 3.5, 33)]
    d2 = Histogram2D(axis=(0, 2), edges=e2, bandwidth=(0.5, 0.5), noise=True, noise_scale=0.0)
    d2.kde = False
    h2 = d2(xs)
    import math
    m2 = h2 / torch.sum(h2) / math.prod([e[1] - e[0] for e in d2.edges])
    _save("ref_hist_hard", x=_np(xs), edges1=_np(edges), hist1=_np(h1), meas1=_np(m1),
          edges2x=_np(e2[0]), edges2y=_np(e2[1]), hist2=_np(h2), meas2=_np(m2))

    # ---- 5. directions (experiments/rec_nd_1d/setup.py:28-37 logic, CPU generator) — re-typed, see oracle.model
    for seed, P in ((2, 25), (0, 100)):
        rng = torch.Generator(device="cpu")
        rng.manual_seed(seed)
        dirs = torch.randn((P, 6), generator=rng, device="cpu")
        dirs = dirs / torch.norm(dirs, dim=1)[:, None]
        _save(f"ref_directions_seed{seed}_P{P}_d6", V=_np(dirs))

    # ---- 6. simulate.forward list structure (25 transforms, one shared Histogram1D)
    rng = torch.Generator(device="cpu").manual_seed(2)
    dirs = torch.randn((25, 6), generator=rng)
    dirs = dirs / torch.norm(dirs, dim=1)[:, None]
    transforms = []
    for direction in dirs:
        Mi = torch.eye(6)
        Mi[0, :] = direction
        transforms.append(LinearTransform(Mi.float()))
    diag = Histogram1D(axis=0, edges=torch.linspace(-4.0, 4.0, 65), bandwidth=0.5)
    diagnostics = [[diag] for _ in transforms]
    xf = randn(2048, 6) * 1.2
    preds = forward(xf, transforms, diagnostics)
    assert len(preds) == 25 and all(len(p) == 1 for p in preds)
    _save("ref_forward_list", x=_np(xf), V=_np(dirs), edges=_np(diag.edges),
          preds=np.stack([_np(p[0]) for p in preds]))

    # ---- 7. kl / mae / mse
    pred = torch.rand(64, generator=g) + 0.01
    targ = torch.rand(64, generator=g)
    targ[::7] = 0.0
    pred2 = torch.rand(32, 32, generator=g) + 0.01
    targ2 = torch.rand(32, 32, generator=g)
    targ2[::5, ::3] = 0.0
    _save("ref_losses", pred=_np(pred), targ=_np(targ), pred2=_np(pred2), targ2=_np(targ2),
          kl=_np(kl_divergence(pred, targ)), mae=_np(mean_absolute_error(pred, targ)),
          mse=_np(mean_square_error(pred, targ)), kl2=_np(kl_divergence(pred2, targ2)),
          mae2=_np(mean_absolute_error(pred2, targ2)), mse2=_np(mean_square_error(pred2, targ2)))

    # ---- 8. MC entropy + Gaussian prior
    xe = randn(4096, 6) * 1.4
    lp = randn(4096) - 8.0
    out = {}
    for s in (1.0, 3.0):
        est = MonteCarloEntropyEstimator(prior=Gaussian(ndim=6, scale=s))
        out[f"H_scale{s}"] = _np(est(xe, lp))
        out[f"prior_logp_scale{s}"] = _np(Gaussian(ndim=6, scale=s).log_prob(xe))
    out["H_noprior"] = _np(MonteCarloEntropyEstimator(prior=None)(xe, lp))
    _save("ref_entropy_mc", x=_np(xe), log_prob=_np(lp), **out)

    # ---- 9. full MENTFlow.loss with an injected generator
    class Injected(GenerativeModel):
        def __init__(self, x, logp):
            super().__init__()
            self.x, self.logp = x, logp

        def sample(self, n):
            return self.x

        def log_prob(self, x):
            return self.logp

        def sample_and_log_prob(self, n):
            return self.x, self.logp

    def loss_case(name, transforms, diagnostics, ndim, n, scale, x_true_scale=1.0):
        xt = randn(20000, ndim) * x_true_scale
        for dd in mf.utils.unravel(diagnostics):
            dd.kde = False
        meas = forward(xt, transforms, diagnostics)
        for dd in mf.utils.unravel(diagnostics):
            dd.kde = True
        for i in range(len(meas)):
            for j in range(len(meas[i])):
                mm, dg = meas[i][j], diagnostics[i][j]
                if mm.ndim == 1:
                    vol = dg.edges[1] - dg.edges[0]
                else:
                    vol = math.prod([e[1] - e[0] for e in dg.edges])
                meas[i][j] = mm / torch.sum(mm) / vol
        xin = (randn(n, ndim) * 1.1)
        lpin = randn(n) - 0.5 * ndim
        res = {}
        for mu in (0.0, 500.0):
            xq = xin.clone().requires_grad_(True)
            lq = lpin.clone().requires_grad_(True)
            prior = Gaussian(ndim=ndim, scale=scale)
            model = MENTFlow(transforms=transforms, diagnostics=diagnostics, measurements=meas,
                             generator=Injected(xq, lq), prior=prior,
                             entropy_estimator=MonteCarloEntropyEstimator(prior=prior),
                             discrepancy_function=kl_divergence, penalty_parameter=mu)
            L, H, D = model.loss(n)
            L.backward()
            tag = f"mu{int(mu)}"
            res[f"L_{tag}"] = _np(L)
            res[f"H_{tag}"] = _np(H)
            res[f"D_{tag}"] = np.array([float(v) for v in D], dtype=np.float32)
            res[f"gx_{tag}"] = _np(xq.grad)
            res[f"glogp_{tag}"] = _np(lq.grad)
        meas_arr = np.stack([_np(m[0]) for m in meas])
        mats = np.stack([_np(t.matrix) for t in transforms])
        _save(name, x=_np(xin), log_prob=_np(lpin), matrices=mats, measurements=meas_arr,
              prior_scale=np.float32(scale), **res)
        return res

    def nd1d(P, seed, xmax, bins=64):
        rng = torch.Generator(device="cpu").manual_seed(seed)
        dirs = torch.randn((P, 6), generator=rng)
        dirs = dirs / torch.norm(dirs, dim=1)[:, None]
        ts = []
        for direction in dirs:
            Mi = torch.eye(6)
            Mi[0, :] = direction
            ts.append(LinearTransform(Mi.float()))
        dg = Histogram1D(axis=0, edges=torch.linspace(-xmax, xmax, bins + 1), bandwidth=0.5)
        return ts, [[dg] for _ in ts]

    ts, dgs = nd1d(25, 2, 4.0)
    loss_case("ref_mentflow_loss_1d_P25", ts, dgs, 6, 2048, 1.0)
    ts, dgs = nd1d(100, 0, 3.5)
    loss_case("ref_mentflow_loss_1d_P100", ts, dgs, 6, 1024, 3.0)

    # 2-D rotations (rec_2d/linear): 7 angles, 85 bins, xmax 3.5
    import numpy as _n
    angles = _n.linspace(0.0, _n.pi, 7, endpoint=False)
    ts = [LinearTransform(mf.simulate.rotation_matrix(a).type(torch.float32)) for a in angles]
    dg = Histogram1D(axis=0, edges=torch.linspace(-3.5, 3.5, 86), bandwidth=0.5)
    loss_case("ref_mentflow_loss_2d_P7", ts, [[dg] for _ in ts], 2, 2048, 1.0)

    # n:2 corner optics (rec_nd_2d/setup.py:38-53), 2-D KDE histograms 48x48
    ts = []
    for i in range(6):
        for j in range(i):
            mats = []
            for k, l in zip((0, 2), (j, i)):
                mm = torch.eye(6)
                mm[k, k] = mm[l, l] = 0.0
                mm[k, l] = mm[l, k] = 1.0
                mats.append(mm.float())
            ts.append(LinearTransform(torch.linalg.multi_dot(mats[::-1])))
    e2 = [torch.linspace(-3.5, 3.5, 49), torch.linspace(-3.5, 3.5, 49)]
    dg2 = Histogram2D(axis=(0, 2), edges=e2, bandwidth=(0.5, 0.5))
    loss_case("ref_mentflow_loss_nd2d_corner15", ts, [[dg2] for _ in ts], 6, 1024, 1.0)

    # ---- 10. ground-truth distributions used by the benchmark configs (host-side data generation)
    from mentflow.distributions import get_distribution
    for name, kws in (("rings", dict(ndim=6, seed=2, decay=0.2)),
                      ("gaussian_mixture", dict(ndim=6, seed=0)),
                      ("swissroll", dict(ndim=2, seed=21))):
        dist = get_distribution(name, **kws)
        xs = dist.sample(20000)
        _save(f"ref_dist_{name}", x=_np(xs[:2048]), mean=_np(xs.mean(0)), std=_np(xs.std(0)),
              n=np.int64(20000))


def main_nonlinear():
    """Fixtures for the non-linear transport (MultipoleTransform / CompositeTransform, rec_2d/nonlinear) and for the
    NN-generator configuration's loss (no entropy term, MAE discrepancy).  Separate RNG stream: running this does not
    change the fixtures written by main()."""
    import math
    mf = _boot()
    sys.modules.setdefault("scipy.special", __import__("scipy.special").special)
    from mentflow.diagnostics import Histogram1D
    from mentflow.simulate import LinearTransform, MultipoleTransform, CompositeTransform, forward, rotation_matrix
    from mentflow.entropy import MonteCarloEntropyEstimator, EmptyEntropyEstimator
    from mentflow.prior import Gaussian
    from mentflow.loss import kl_divergence, mean_absolute_error
    from mentflow.core import MENTFlow
    from mentflow.generate import GenerativeModel

    g = torch.Generator().manual_seed(20250621)
    randn = lambda *s: torch.randn(*s, generator=g)

    # ---- 11. the kick itself: values and vector-Jacobian products, d = 2, 4 and 6
    out = {}
    for d in (2, 4, 6):
        x = randn(512, d) * 1.2
        w = randn(512, d)
        out[f"x_d{d}"], out[f"w_d{d}"] = _np(x), _np(w)
        for order in (3, 4, 5):
            for skew in (False, True):
                xq = x.clone().requires_grad_(True)
                t = MultipoleTransform(order=order, strength=0.7 * order, skew=skew)
                u = t(xq)
                (u * w).sum().backward()
                tag = f"d{d}_o{order}_s{int(skew)}"
                out[f"u_{tag}"], out[f"gx_{tag}"] = _np(u).copy(), _np(xq.grad)
                # the reference's reverse_momentum (transform.py:18-21) flips momenta IN PLACE: hand it a copy
                out[f"inv_{tag}"] = _np(t.inverse(u.detach().clone()))
    _save("ref_multipole", **out)

    # ---- 12. MENTFlow.loss through CompositeTransform(multipole, rotation): rec_2d/nonlinear (4 strengths, order 3)
    class Injected(GenerativeModel):
        def __init__(self, x, logp):
            super().__init__()
            self.x, self.logp = x, logp

        def sample(self, n):
            return self.x

        def log_prob(self, x):
            return self.logp

        def sample_and_log_prob(self, n):
            return self.x, self.logp

    strengths = np.linspace(-1.5, 1.5, 4)
    ts = []
    for strength in strengths:
        rot = LinearTransform(rotation_matrix(np.radians(90.0)).type(torch.float32))
        ts.append(CompositeTransform(MultipoleTransform(order=3, strength=strength), rot))
    dg = Histogram1D(axis=0, edges=torch.linspace(-4.5, 4.5, 86), bandwidth=0.5)
    dgs = [[dg] for _ in ts]
    xt = randn(20000, 2)
    dg.kde = False
    meas = forward(xt, ts, dgs)
    dg.kde = True
    for i in range(len(meas)):
        mm = meas[i][0]
        meas[i][0] = mm / torch.sum(mm) / (dg.edges[1] - dg.edges[0])
    n = 2048
    xin, lpin = randn(n, 2) * 1.1, randn(n) - 1.0
    res = {}
    for tag, ent, disc, mu in (("flow", "mc", kl_divergence, 500.0), ("nn", "none", mean_absolute_error, 500.0)):
        xq = xin.clone().requires_grad_(True)
        lq = lpin.clone().requires_grad_(True)
        prior = Gaussian(ndim=2, scale=1.0)
        est = MonteCarloEntropyEstimator(prior=prior) if ent == "mc" else EmptyEntropyEstimator()
        model = MENTFlow(transforms=ts, diagnostics=dgs, measurements=meas,
                         generator=Injected(xq, lq if ent == "mc" else None), prior=prior,
                         entropy_estimator=est, discrepancy_function=disc, penalty_parameter=mu)
        L, H, D = model.loss(n)
        L.backward()
        res[f"L_{tag}"], res[f"H_{tag}"] = _np(L), np.float32(float(H))
        res[f"D_{tag}"] = np.array([float(v) for v in D], dtype=np.float32)
        res[f"gx_{tag}"] = _np(xq.grad)
        if ent == "mc":
            res[f"glogp_{tag}"] = _np(lq.grad)
    _save("ref_mentflow_loss_2d_nonlinear", x=_np(xin), log_prob=_np(lpin), strengths=strengths.astype(np.float64),
          angle_deg=np.float64(90.0), order=np.int64(3), edges=_np(dg.edges),
          measurements=np.stack([_np(m[0]) for m in meas]), **res)


if __name__ == "__main__":
    if "--nonlinear" in sys.argv:
        main_nonlinear()
    else:
        main()
