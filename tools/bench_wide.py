"""Timing of the wide conditioner family (hidden_units 65 .. 128 and / or 8 .. 16 features; mentflow_amd/csrc/flow_wide.hip) on one
GPU: the C4 problem (rec_nd_1d gaussian_mixture d = 6, 100 projections x 64 bins, NSF 5 layers, 20 bins) with another conditioner
shape, one MENTFlow.loss() + backward + AdamW step per iteration — the step bench.py times, just not the headline configuration.

    python tools/bench_wide.py [--hidden-units 128] [--hidden-layers 3] [--ndim 6] [--per-gpu 1048576] [--steps 10]

Prints one JSON line: ms per step, particle-samples/s, the per-kernel split from the library's HIP-event profile and the flow
kernels' dense-equivalent fp32 FLOP rate: F = 2 (d h + (L-1) h^2 + h q d) per particle and layer for the forward; the per-tile
backward kernel runs the transposed chains (F; 2 F when it also recomputes the conditioner: MENTFLOW_ACT_LEVEL=0), the contraction
kernel forms the parameter gradients (F).  Dense-equivalent: the kernels skip the all-zero 32 x 32 blocks of the autoregressive masks."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import mentflow_amd as mf  # noqa: E402,F401
from mentflow_amd import _lib  # noqa: E402
from mentflow_amd.harness import build_problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden-units", type=int, default=128)
    ap.add_argument("--hidden-layers", type=int, default=3)
    ap.add_argument("--ndim", type=int, default=6)
    ap.add_argument("--per-gpu", type=int, default=1_048_576)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num", type=int, default=100)
    ap.add_argument("--bwd-chunk", type=int, default=None, help="particles per backward chunk (scratch = 4.6 KB per particle at 128 units)")
    ap.add_argument("--lib", default=None, help="another build of the library (A/B runs: tools/build_variant.sh)")
    args = ap.parse_args()
    if args.lib:
        _lib.use_library(os.path.abspath(args.lib))
    dev = torch.device("cuda", 0)
    prob = build_problem(device=dev, penalty_parameter=500.0, ndim=args.ndim, num=args.num, bins=64, xmax=3.5, seed=0, transforms=5,
                         prior_scale=3.0, dist_name="gaussian_mixture", optics="nd_1d", hidden_units=args.hidden_units,
                         hidden_layers=args.hidden_layers, meas_samples=200_000)
    model = prob.model
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
    n = args.per_gpu
    if args.bwd_chunk:
        model.generator.spec().bwd_chunk = args.bwd_chunk

    def step():
        opt.zero_grad()
        L, H, D = model.loss(n)
        L.backward()
        opt.step()
        return L

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = _lib.prof_report()
    _lib.prof_enable(False)
    d, h, Lh, q, T = args.ndim, args.hidden_units, args.hidden_layers, 59, 5
    flops = 2 * (d * h + (Lh - 1) * h * h + h * q * d)
    level = model.generator.spec().resolve_act_level(n, dev)
    out = {"wide": bool(model.generator.wide), "activation_handoff": int(level), "ndim": d, "hidden_units": h, "hidden_layers": Lh, "particles": n,
           "ms_per_step": 1e3 * el / args.steps, "particle_samples_per_s": n * args.steps / el, "final_loss": float(L.detach()),
           "kernels_ms_per_step": {k: ms / args.steps for k, (ms, cnt) in prof.items() if cnt},
           "launches_per_step": {k: cnt / args.steps for k, (ms, cnt) in prof.items() if cnt}}
    ks = out["kernels_ms_per_step"]
    if "flow_layer_fwd" in ks:
        out["fwd_dense_tflops"] = flops * n * T / (ks["flow_layer_fwd"] * 1e-3) / 1e12
    if "flow_layer_bwd" in ks:
        out["bwd_dense_tflops"] = (1 if (level > 0 and model.generator.wide) else 2) * flops * n * T / (ks["flow_layer_bwd"] * 1e-3) / 1e12
    if "outer_accum" in ks:
        out["outer_accum_dense_tflops"] = flops * n * T / (ks["outer_accum"] * 1e-3) / 1e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
