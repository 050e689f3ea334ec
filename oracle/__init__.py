"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (PyTorch eager, fp32 or fp64) restatement of the reference algorithm for the
MENT-Flow hot path (austin-hoover/ment-flow @ 2025-06-20):

    flow sample + log-det  ->  linear projections  ->  KDE histograms
                           ->  Monte-Carlo entropy + KL data mismatch  (one MENTFlow.loss()).

Who may import this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — and there only as the checker / timed CPU baseline,
never as the thing shipped.  ``mentflow_amd`` (the product) never imports it.

Pinning status
--------------
* ``oracle.kde``, ``oracle.model`` (LinearTransform / simulate.forward / Histogram1D/2D /
  MonteCarloEntropyEstimator / Gaussian prior / kl_divergence / MENTFlow.loss):
  **pinned** — checked against outputs of the reference's own code imported in the build
  container (``oracle/gen_golden.py`` -> ``tests/golden/ref_*.npz``, ``tests/test_oracle_golden.py``).
* ``oracle.flow`` (masked-MLP conditioner, rational-quadratic-spline / affine autoregressive
  transforms): the arithmetic lives in third-party ``zuko==1.3.1`` (reference
  ``pyproject.toml:11``), which is NOT vendored under /root/reference and not installed.
  The reference holds no test or golden vector at that boundary.  **parity unpinned** for
  this module: it restates zuko's published algorithm and is
  validated by properties only (invertibility, log-det vs autograd Jacobian, autoregressive
  mask structure, identity outside the spline domain, fp64 gradcheck).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
