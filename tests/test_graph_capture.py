"""hipGraph capture of the training step (GPU): a replayed step must equal the eager step bit for bit on an injected
base draw, keep training (fresh particles per replay)."""
import copy
import ctypes
import time

import pytest
import torch

from conftest import set_bwd_variant

import mentflow_amd as mf
from mentflow_amd.harness import build_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,fused", [(25_000, "0"), (25_000, "1"), (40_000, "1")])   # two-kernel / fused backward
def test_graphed_step_equals_eager_and_trains(n, fused, monkeypatch):
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)                 # FIRST: the variant switch is per-library state
    set_bwd_variant(monkeypatch, fused)
    # the two-kernel variant really is the one that runs: it is the only one that needs hand-off scratch
    order = (ctypes.c_int32 * 6)(*range(6))
    assert (_lib.get_lib().mf_flow_bwd_scratch_floats(n, 6, 3, order) > 0) == (fused == "0")
    dev = torch.device("cuda", 0)
    prob = build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=dev,
                         dist_name="rings", meas_samples=200_000, penalty_parameter=100.0)
    model = prob.model
    state0 = copy.deepcopy(model.state_dict())
    torch.manual_seed(0)
    z = torch.randn(n, 6, device=dev)

    def run(graphed: bool):
        model.load_state_dict(state0)
        model.generator.inject_z = z
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, capturable=True)
        if graphed:
            # the 3 eager warm-up steps that graph capture needs are undone by GraphedTrainStep (parameters and AdamW
            # state restored in place): the 3 replays are training steps 1-3, exactly like the 3 eager steps
            g = mf.graph.GraphedTrainStep(model, opt, n, warmup=3)
            outs = [tuple(t.clone() for t in g.step()) for _ in range(3)]
        else:
            outs = []
            for it in range(3):
                opt.zero_grad(set_to_none=False)
                L, H, D = model.loss(n)
                L.backward()
                opt.step()
                outs.append((L.detach().clone(), H.detach().clone(), torch.stack(D).detach().mean()))
        params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        return outs, params

    eager, pe = run(False)
    graphed, pg = run(True)
    # every kernel on the path is deterministic (fixed-point histograms with integer flushes, slab-reduced parameter
    # gradients: no float atomics anywhere): a replayed step reproduces the eager step BIT FOR BIT, and so do the
    # parameters after 3 AdamW steps
    for a, b in zip(eager, graphed):
        for u, v in zip(a, b):
            assert torch.equal(u, v), (u, v)
    assert torch.equal(pe, pg), float((pe - pg).abs().max())
    # fresh particles per replay + speed at the reference batch size
    model.generator.inject_z = None
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, capturable=True)
    g = mf.graph.GraphedTrainStep(model, opt, n)
    l1 = float(g.step()[0]); l2 = float(g.step()[0])
    assert l1 != l2
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        g.step()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 50
    opt2 = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        opt2.zero_grad(); L, H, D = model.loss(n); L.backward(); opt2.step()
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 50
    print(f"\n{n}-particle step: eager {t_eager*1e3:.2f} ms, graph replay {t_graph*1e3:.2f} ms")
    # timing is printed, not asserted: a wall-clock gate inside a correctness test goes red on a noisy box and, under -x,
    # hides everything collected after it (the numbers live in profiles/ and DESIGN.md)


def test_graphed_trainer_matches_eager_trainer_and_undo():
    """Trainer(graphed=True): the penalty-method loop replayed from a per-epoch hipGraph must log the same (L, H, D)
    history as the eager Trainer on an injected base draw (bitwise: every kernel is deterministic), across a penalty
    update; GraphedTrainStep.undo_last_step restores parameters and optimizer state exactly."""
    from mentflow_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda", 0)
    n = 25_000
    hist = {}
    finals = {}
    for graphed in (False, True):
        prob = build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=dev,
                             dist_name="rings", meas_samples=100_000)
        model = prob.model
        torch.manual_seed(0)
        model.generator.inject_z = torch.randn(n, 6, device=dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, capturable=True)
        tr = mf.train.Trainer(model, opt, None, verbose=False, graphed=graphed)
        tr.train(epochs=2, iterations=4, batch_size=n, rtol=-1, atol=-1, dmax=1e-12, penalty_start=10.0, penalty_step=20.0,
                 penalty_scale=1.5, eval_batch_size=n)
        hist[graphed] = tr.history
        finals[graphed] = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    for key in ("L", "H", "D_norm", "penalty"):
        assert hist[False][key] == hist[True][key], key
    assert torch.equal(finals[False], finals[True])

    # undo: one replay, then back to exactly the previous parameters / AdamW state
    g = mf.graph.GraphedTrainStep(model, opt, n, guard=True)
    before = [t.clone() for t in g._state]
    g.step()
    torch.cuda.synchronize()
    assert any(not torch.equal(a, b) for a, b in zip(before, g._state))
    g.undo_last_step()
    assert all(torch.equal(a, b) for a, b in zip(before, g._state))


def test_graphed_trainer_with_the_learning_rate_on_the_device():
    """Trainer(graphed=True, lr_on_device=True): a scheduler that moves the rate costs no re-capture (the captured AdamW reads
    the rate from a device tensor that the Trainer refreshes after every scheduler step), while the default float rate
    re-captures on every change.  The reference's scheduler API: ReduceLROnPlateau.step(loss) (train.py:214)."""
    from mentflow_amd import _lib
    from mentflow_amd import graph as mfgraph
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda", 0)
    n = 25_000
    hist, captures = {}, {}
    real_recapture = mfgraph.GraphedTrainStep.recapture
    for on_device in (False, True):
        prob = build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=dev,
                             dist_name="rings", meas_samples=100_000)
        model = prob.model
        torch.manual_seed(0)
        model.generator.inject_z = torch.randn(n, 6, device=dev)
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.0, capturable=True)
        # mode="max" on a decreasing loss + patience 0: the rate drops by 0.7 at (almost) every step
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.7, patience=0, min_lr=1e-5)
        count = [0]

        def counting(self, _c=count):
            _c[0] += 1
            return real_recapture(self)

        mfgraph.GraphedTrainStep.recapture = counting
        try:
            tr = mf.train.Trainer(model, opt, sched, verbose=False, graphed=True, lr_on_device=on_device)
            tr.train(epochs=1, iterations=6, batch_size=n, rtol=-1, atol=-1, dmax=1e-12, penalty_start=10.0, eval_batch_size=n)
        finally:
            mfgraph.GraphedTrainStep.recapture = real_recapture
        hist[on_device], captures[on_device] = tr.history, count[0]
    lrs = hist[True]["learning_rate"]
    assert captures[True] == 1, captures                        # the construction-time capture only
    assert captures[False] >= 4, captures                       # float rate: one re-capture per change
    assert lrs[0] == pytest.approx(2e-3) and lrs[-1] < 0.5 * lrs[0] and all(a >= b for a, b in zip(lrs, lrs[1:])), lrs
    assert hist[False]["learning_rate"] == pytest.approx(lrs, rel=1e-6)
    Ls = hist[True]["L"]
    assert all(v == v for v in Ls) and Ls[-1] < Ls[0]
    # same training up to the last bits of AdamW's rate arithmetic (fp32 tensor rate against a folded python float)
    assert hist[False]["L"] == pytest.approx(Ls, rel=1e-4)
