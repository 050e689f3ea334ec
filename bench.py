#!/usr/bin/env python
"""bench.py — particle-samples/s of one MENT-Flow training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is what mentflow/train/train.py:164-169 does per iteration: optimizer.zero_grad(); model.loss(batch);
loss.backward(); AdamW.step() — on synthetic data of BASELINE.json's headline configuration (C4: 6-D, 100 random
1-D projections, 64 bins, xmax 3.5, NSF flow 5x[3x64], K=20, gaussian-mixture ground truth, prior scale 3,
penalty 500), 2 097 152 particles per GPU (weak scaling: the 16 M-particle batch of C4 over 8 GPUs).
Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (HIP-event timed inside
the timed region) and the CPU baseline (the oracle restatement of the reference timed on this host's cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mentflow_amd as mf                                   # noqa: E402
from mentflow_amd import _lib                               # noqa: E402
from mentflow_amd import dist as mfdist                     # noqa: E402
from mentflow_amd.harness import build_problem              # noqa: E402

# dense-contraction FLOPs of the conditioner, per particle and per flow layer:
#   2 * (d*h + 2*h^2 + h*q*d)  with h = 64, q = 3K-1 = 59
def layer_flops(d: int, h: int = 64, hidden_layers: int = 3, q: int = 59) -> int:
    return 2 * (d * h + (hidden_layers - 1) * h * h + h * q * d)


PEAK_MFMA_F32 = 157.3          # TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (fp32-input MFMA, dense)
PEAK_HBM = 8000.0              # GB/s spec

WORKLOADS = {
    # name: build_problem kwargs + per-GPU batch
    "c4": dict(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, dist_name="gaussian_mixture",
               optics="nd_1d", per_gpu=2_097_152,
               desc="rec_nd_1d gaussian_mixture d=6, 100 linear 1-D projections x 64 bins, NSF 5x[3x64] K=20"),
    "c3": dict(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, dist_name="rings",
               optics="nd_1d", per_gpu=4_194_304,
               desc="rec_nd_1d rings d=6, 25 linear 1-D projections x 64 bins, NSF 5x[3x64] K=20"),
    "c1": dict(ndim=2, num=7, bins=85, xmax=3.5, seed=21, transforms=5, prior_scale=1.0, dist_name="swissroll",
               optics="2d_linear", gen_name="maf", per_gpu=50_000,
               desc="rec_2d/linear swissroll d=2, 7 projections x 85 bins, MAF (affine) 5x[3x64], the reference's 50k batch"),
    "c2": dict(ndim=2, num=7, bins=85, xmax=3.5, seed=21, transforms=5, prior_scale=1.0, dist_name="swissroll",
               optics="2d_linear", per_gpu=1_048_576,
               desc="rec_2d/linear swissroll d=2, 7 projections x 85 bins, NSF 5x[3x64] K=20"),
    "c5": dict(ndim=6, num=100, bins=85, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, dist_name="gaussian_mixture",
               optics="nd_2d_random", per_gpu=2_097_152,
               desc="rec_nd_2d d=6, 100 2-D projections x 85x85 bins, NSF 5x[3x64] K=20"),
}


def host_cores() -> int:
    """Cores this process may actually use: the scheduler affinity, capped at the GPU box's per-GPU CPU share (16) —
    os.cpu_count() reports every core of the host and oversubscribes the OpenMP pool."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(prob, budget_s: float = 20.0, n_cpu: int = 25_000):
    """The oracle (eager dense restatement of the reference semantics) timed on this host: same problem, the
    reference's own batch size (experiments/rec_nd_1d/run_rings.sh:21), zero_grad + loss + backward + AdamW."""
    from oracle import model as om
    from oracle.harness import oracle_problem
    torch.set_num_threads(host_cores())
    spec, transforms, diagnostics, measurements, prior, disc = oracle_problem(prob)
    params = spec.parameters()
    for p in params:
        p.requires_grad_(True)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.0)
    gen = torch.Generator().manual_seed(1234)
    mu = float(prob.model.penalty_parameter)

    def step():
        opt.zero_grad()
        z = torch.randn(n_cpu, spec.features, generator=gen)
        L, _, _, _, _ = om.train_step_loss(z, spec, transforms, diagnostics, measurements, prior, mu, disc)
        L.backward()
        opt.step()

    t0 = time.perf_counter()
    step()                                                   # warm-up (allocator, thread pool)
    log(f"cpu_baseline: warm-up step {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
    t0 = time.perf_counter()
    steps = 0
    while True:
        step()
        steps += 1
        el = time.perf_counter() - t0
        log(f"cpu_baseline: {steps} steps, {el:.1f} s")
        if el > budget_s or steps >= 50:
            break
    out = {"value": n_cpu * steps / el, "unit": "particle-samples/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{steps} train steps of {n_cpu} particles (reference batch size) on the same workload, "
                     f"{el:.1f} s of CPU work; oracle = eager dense PyTorch restatement of the reference"}
    out["parity"] = parity_gate(prob)
    return out


def parity_gate(prob, n: int = 36864):
    """Same leg as the CPU baseline (the only place bench.py may touch the oracle): one loss + backward of the SAME model
    on the GPU and in the oracle from one injected base draw (36 864 particles: the fused backward's size range), so
    that the bench line carries the evidence that what was timed computes what the reference computes."""
    from oracle.harness import oracle_step
    model = prob.model
    gen = model.generator
    z = torch.randn(n, gen.features, generator=torch.Generator().manual_seed(4321))
    dev = next(model.parameters()).device
    saved = gen.inject_z
    gen.inject_z = z.to(dev)
    model.zero_grad()
    L, H, D = model.loss(n)
    L.backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    gen.inject_z = saved
    model.zero_grad()
    Lo, Ho, Do, go = oracle_step(prob, z)
    res = {"particles": n, "L_abs_err": abs(float(L) - float(Lo)), "H_abs_err": abs(float(H) - float(Ho)),
           "D_max_abs_err": float((torch.stack(D).detach().cpu() - torch.stack(Do)).abs().max()),
           "grad_max_err_over_max_grad": float((g - go).abs().max() / go.abs().max()), "L": float(Lo)}
    mu = float(model.penalty_parameter)
    res["ok"] = bool(res["L_abs_err"] < 1e-4 + mu * 2e-6 + 2e-5 * abs(float(Lo)) and res["H_abs_err"] < 1e-4
                     and res["grad_max_err_over_max_grad"] < 2e-3)
    log("parity gate: " + json.dumps(res))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--per-gpu", type=int, default=None, help="particles per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--bwd-chunk", type=int, default=None, help="particles per flow-backward chunk (tuning)")
    args = ap.parse_args()

    device = mfdist.init_from_env()
    world = mfdist.world_size()
    rank = mfdist.rank()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: mentflow_amd has no CPU path")

    w = dict(WORKLOADS[args.workload])
    per_gpu = args.per_gpu or w.pop("per_gpu")
    w.pop("per_gpu", None)
    desc = w.pop("desc")
    prob = build_problem(device=device, penalty_parameter=500.0, **w)       # same seed on every rank: same weights
    model = prob.model
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0)   # experiments/setup.py:166-170
    torch.manual_seed(1234 + rank)                                           # every rank draws its own particles
    if args.bwd_chunk:
        model.generator.spec().bwd_chunk = args.bwd_chunk
    global_batch = per_gpu * world

    def step():
        opt.zero_grad()
        L, H, D = model.loss(global_batch)
        L.backward()
        opt.step()
        return L

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    mfdist.barrier()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L = step()
    torch.cuda.synchronize()
    mfdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = _lib.prof_report()
    _lib.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])
    final_loss = float(L.detach())

    if rank == 0:
        d = w["ndim"]
        lf = layer_flops(d, q=2 if w.get("gen_name") == "maf" else 59)      # MAF: shift + scale per feature
        T = w["transforms"]
        value = global_batch * args.steps / elapsed
        # dominant kernel by summed HIP-event time inside the timed region
        flow_kernels = {k: v for k, v in prof.items() if v[1] > 0}
        dom = max(flow_kernels, key=lambda k: flow_kernels[k][0])
        dom_ms, dom_cnt = flow_kernels[dom]
        # particles one launch of the dominant kernel processes (backward runs in chunks)
        launches_per_step = dom_cnt / args.steps
        per_launch_particles = per_gpu * (T if dom in ("flow_layer_fwd", "flow_layer_bwd", "outer_accum") else 1) / launches_per_step
        # the fused backward kernel (no outer_accum launches) does backward-data AND the parameter gradients: 2 F_layer
        fused_bwd = "outer_accum" not in flow_kernels
        alg_flops = {"flow_layer_fwd": lf, "flow_layer_bwd": 2 * lf if fused_bwd else lf, "outer_accum": lf}.get(dom)
        roof = {"kernel": dom + (" (fused: backward-data + parameter gradients)" if dom == "flow_layer_bwd" and fused_bwd else ""),
                "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_MFMA_F32, "traffic": None}
        if alg_flops is not None:
            roof["achieved"] = alg_flops * per_launch_particles / (dom_ms / dom_cnt * 1e-3) / 1e12
        else:   # a KDE kernel dominates: HBM-bound byte work, algorithmic bytes = particle rows read (+ written)
            nbytes = (4 * d) * per_launch_particles * (2 if dom.endswith("bwd") else 1)
            roof.update(bound="hbm", unit="GB/s", peak=PEAK_HBM)
            roof["achieved"] = nbytes / (dom_ms / dom_cnt * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["avg_launch_ms"] = dom_ms / dom_cnt
        roof["launches"] = dom_cnt
        roof["algorithmic_per_launch"] = (f"{alg_flops} FLOP/particle x {int(per_launch_particles)} particles"
                                          if alg_flops else "particle rows")
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                roof["traffic"] = json.load(open(traffic_file)).get(dom)
            except Exception:
                pass
        out = {
            "metric": "particle-samples/sec per MENT-Flow train step (6D, 100 proj)" if args.workload == "c4"
                      else f"particle-samples/sec per MENT-Flow train step ({args.workload})",
            "value": value, "unit": "particle-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "global_batch": global_batch, "per_gpu_batch": per_gpu,
                       "parallelism": f"dp{world} (particle batch sharded; 2 all-reduces/step)",
                       "step": "zero_grad + MENTFlow.loss + backward + AdamW.step", "final_loss": final_loss},
            "roofline": roof,
            "step_mfma_frac": 3 * T * lf * (value / world) / (PEAK_MFMA_F32 * 1e12),
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1] > 0},
        }
        log("gpu leg: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline", "kernel_ms_per_step")}))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_budget)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    mfdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
