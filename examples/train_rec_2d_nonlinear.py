#!/usr/bin/env python
"""2-D reconstruction through a NON-LINEAR transport on one MI355X — the hydra-free equivalent of
`experiments/rec_2d/nonlinear/train_flow.py` / `train_nn.py` (configs rec_2d_nonlinear_flow.yaml / _nn.yaml: rings, 4
measurements, thin sextupole kick of strength -1.5 ... +1.5 followed by a 90-degree rotation, 85 bins, xmax 4.5).

    python examples/train_rec_2d_nonlinear.py --gen nsf        # flow: MC entropy + KL
    python examples/train_rec_2d_nonlinear.py --gen nn         # plain network: no entropy term, MAE
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
    ap.add_argument("--gen", default="nsf", choices=["nsf", "nn"])
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=40000)
    ap.add_argument("--seed", type=int, default=21)
    args = ap.parse_args()

    dev = torch.device("cuda", 0)
    nn_gen = args.gen == "nn"
    prob = build_problem(ndim=2, num=4, bins=85, xmax=4.5, seed=args.seed, transforms=5, prior_scale=1.0, device=dev,
                         dist_name="rings", meas_samples=1_000_000, optics="2d_nonlinear", gen_name=args.gen,
                         hidden_layers=3, hidden_units=50 if nn_gen else 64, discrepancy="mae" if nn_gen else "kld")
    model = prob.model
    torch.manual_seed(args.seed)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3 if not nn_gen else 1e-2, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, min_lr=1e-4, patience=400, factor=0.1)
    trainer = mf.train.Trainer(model, opt, sched, verbose=True)
    t0 = time.time()
    trainer.train(epochs=args.epochs, iterations=args.iters, batch_size=args.batch_size, rtol=-1, atol=-1, dmax=1e-5,
                  penalty_start=500.0 if nn_gen else 0.0, penalty_step=0.0 if nn_gen else 50.0,
                  penalty_scale=1.0 if nn_gen else 1.5, eval_batch_size=100000)
    torch.cuda.synchronize()
    dt = time.time() - t0
    h = trainer.history
    steps = len(h["L"])
    print(f"{args.gen}: {steps} steps of {args.batch_size} particles in {dt:.1f} s; D first {h['D_norm'][0]:.3e} -> last "
          f"{h['D_norm'][-1]:.3e}")
    with torch.no_grad():
        x = model.sample(1_000_000)
        preds = mf.simulate.forward(x, model.transforms, model.diagnostics)
        D = torch.stack([mf.loss.mean_absolute_error(p[0], m[0]) for p, m in zip(preds, model.measurements)])
    print(f"final mean |pred - meas| over {len(preds)} non-linear views with 1e6 particles: {float(D.mean()):.3e}")


if __name__ == "__main__":
    main()
