#!/usr/bin/env python
"""End-to-end MENT-Flow reconstruction on one MI355X — the hydra-free equivalent of
`experiments/rec_nd_1d/train_flow.py ndim=6 seed=2 meas.num=25 meas.bins=64 meas.xmax=4.0 dist.name=rings +dist.decay=0.2
model.prior_scale=1.0 gen.transforms=5 train.batch_size=25000` (experiments/rec_nd_1d/run_rings.sh:33-44), with the
penalty schedule of experiments/config/rec_nd_1d_flow.yaml (penalty 0 -> *1.5 + 50 per epoch, dmax 1e-4).

    python examples/train_rec_nd_1d.py --epochs 3 --iters 100 --batch-size 25000
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mentflow_amd as mf                                    # noqa: E402
from mentflow_amd.harness import build_problem               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndim", type=int, default=6)
    ap.add_argument("--meas-num", type=int, default=25)
    ap.add_argument("--bins", type=int, default=64)
    ap.add_argument("--xmax", type=float, default=4.0)
    ap.add_argument("--dist", default="rings")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=25000)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--hidden-units", type=int, default=64, help="config/gen/flow.yaml:3; above 64 the wide kernel family runs")
    ap.add_argument("--hidden-layers", type=int, default=3)
    ap.add_argument("--graphed", action="store_true",
                    help="replay each iteration from a per-epoch hipGraph (mentflow_amd.graph) with a fused AdamW: the "
                         "launch-bound 25 000-particle regime, 0.77 ms instead of 1.2 ms per step")
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    prob = build_problem(ndim=args.ndim, num=args.meas_num, bins=args.bins, xmax=args.xmax, seed=args.seed, transforms=5,
                         prior_scale=1.0, device=dev, dist_name=args.dist, meas_samples=1_000_000, hidden_units=args.hidden_units,
                         hidden_layers=args.hidden_layers)
    model = prob.model
    torch.manual_seed(args.seed)                                               # experiments/setup.py:163-164
    # setup.py:166-170; graphed: capturable + fused (the capturable foreach AdamW launches ~130 tiny kernels per step)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=0.0,
                            **(dict(capturable=True, fused=True) if args.graphed else {}))
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, min_lr=args.lr, patience=400, factor=0.1)
    trainer = mf.train.Trainer(model, opt, sched, verbose=True, graphed=args.graphed)
    t0 = time.time()
    trainer.train(epochs=args.epochs, iterations=args.iters, batch_size=args.batch_size, rtol=-1, atol=-1, dmax=1e-4,
                  penalty_start=0.0, penalty_step=50.0, penalty_scale=1.5, eval_batch_size=100000)
    torch.cuda.synchronize()
    dt = time.time() - t0
    h = trainer.history
    steps = len(h["L"])
    print(f"{steps} training steps of {args.batch_size} particles in {dt:.1f} s "
          f"({steps * args.batch_size / dt:.3e} particle-samples/s incl. per-epoch 100k evaluation and host logging)")
    for e in range(args.epochs):
        idx = [i for i, ep in enumerate(h["epoch"]) if ep == e]
        print(f"epoch {e}: penalty {h['penalty'][idx[0]]:.1f}  D first {h['D_norm'][idx[0]]:.3e} -> last {h['D_norm'][idx[-1]]:.3e}"
              f"   H last {h['H'][idx[-1]]:.4f}")
    with torch.no_grad():
        x = model.sample(1_000_000)                              # mentflow/train/plot.py:373-382 sized sample
        preds = mf.simulate.forward(x, model.transforms, model.diagnostics)
        D = torch.stack([mf.loss.kl_divergence(p[0], m[0]) for p, m in zip(preds, model.measurements)])
    print(f"final mean KL over {len(preds)} projections with 1e6 particles: {float(D.mean()):.3e}; sample std {x.std(0).tolist()}")


if __name__ == "__main__":
    main()
