#!/usr/bin/env python
"""bench.py — particle-samples/s of one MENT-Flow training step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

With ``--gpus N > 1`` and no WORLD_SIZE in the environment the script starts its N ranks itself: the parent is a pure
launcher (no torch import, no HIP call) that spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
forwards rank 0's JSON line and exits non-zero if any rank fails.  Under ``torch.distributed.run`` the same ranks are
started by the external launcher instead.

A "step" is what mentflow/train/train.py:164-169 does per iteration: optimizer.zero_grad(); model.loss(batch);
loss.backward(); AdamW.step() — on synthetic data of BASELINE.json's headline configuration (C4: 6-D, 100 random
1-D projections, 64 bins, xmax 3.5, NSF flow 5x[3x64], K=20, gaussian-mixture ground truth, prior scale 3,
penalty 500).  ``--scaling weak`` (default): 2 097 152 particles per GPU (C4's 16 M-particle batch over 8 GPUs);
``--scaling strong``: C4's global 16 777 216-particle batch divided over the N ranks (N = 1: all of it on one GPU).
The timed region (K steps between barrier + synchronize) is repeated ``--repeats`` times; the JSON line reports the
MEDIAN region (max over ranks of each region first) and the spread.  Rank 0 prints ONE JSON line with the throughput, the
roofline of the dominant kernel (HIP-event timed inside the timed regions) and — at N = 1 — the CPU baseline (the oracle
restatement of the reference timed on this host's cores).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_MFMA_F32 = 157.3          # TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (fp32-input MFMA, dense)
PEAK_HBM = 8000.0              # GB/s spec
# 64-bit LDS atomic adds a CU sustains per clock with enough waves in flight, whatever the address pattern
# (tools/ubench_lds_atomics2.hip on MI355X, profiles/r02_ubench_lds_atomics2.txt: 3.8-3.9 lane-ops/clk/CU): the KDE
# forward kernels are bound by this, not by HBM
LDS_ATOMIC_U64_PER_CLK_CU = 3.9
CLOCK_GHZ = 2.4
NUM_CU = 256
# KDE backward kernels: bound by vector-instruction ISSUE, not by HBM (76 GB/s of row traffic).  Instruction mix of the inner
# loop body per (particle, projection) and lane, counted in the gfx950 ISA by tools/isa_loop_mix.py
# (profiles/r04_kde_bwd_instruction_mix.txt): (full-rate VALU, transcendental, LDS).  A wave64 instruction holds its SIMD-32 for
# 2 cycles, a transcendental one for 8 (/opt/skills/guides/MI355X_MICROARCH.md, per-instruction cycle constants); the LDS reads
# (2 cycles per ds_read_b32 on the CU's one LDS) stay under the VALU time of the four SIMDs and are reported, not booked.
# 1-D: the INTERIOR path (whole window inside the grid: all but the outermost 4 bins) = 19 preamble + 59 window + 8 gradient-row
# instructions; the general (edge) path issues 170.
KDE_BWD_MIX = {"kde1d_bwd": (86, 9, 12), "kde2d_bwd": (500, 18, 87)}
VALU_CYCLES, TRANS_CYCLES, SIMDS = 2, 8, 4 * NUM_CU

WORKLOADS = {
    # name: build_problem kwargs + per-GPU batch (weak scaling) + global batch (strong scaling)
    "c4": dict(ndim=6, num=100, bins=64, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, dist_name="gaussian_mixture",
               optics="nd_1d", per_gpu=2_097_152, global_batch=16_777_216,
               desc="rec_nd_1d gaussian_mixture d=6, 100 linear 1-D projections x 64 bins, NSF 5x[3x64] K=20"),
    "c3": dict(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, dist_name="rings",
               optics="nd_1d", per_gpu=4_194_304, global_batch=4_194_304,
               desc="rec_nd_1d rings d=6, 25 linear 1-D projections x 64 bins, NSF 5x[3x64] K=20"),
    "c1": dict(ndim=2, num=7, bins=85, xmax=3.5, seed=21, transforms=5, prior_scale=1.0, dist_name="swissroll",
               optics="2d_linear", gen_name="maf", per_gpu=50_000, global_batch=50_000,
               desc="rec_2d/linear swissroll d=2, 7 projections x 85 bins, MAF (affine) 5x[3x64], the reference's 50k batch"),
    "c2": dict(ndim=2, num=7, bins=85, xmax=3.5, seed=21, transforms=5, prior_scale=1.0, dist_name="swissroll",
               optics="2d_linear", per_gpu=1_048_576, global_batch=1_048_576,
               desc="rec_2d/linear swissroll d=2, 7 projections x 85 bins, NSF 5x[3x64] K=20"),
    "c5": dict(ndim=6, num=100, bins=85, xmax=3.5, seed=0, transforms=5, prior_scale=3.0, dist_name="gaussian_mixture",
               optics="nd_2d_random", per_gpu=2_097_152, global_batch=16_777_216,
               desc="rec_nd_2d d=6, 100 2-D projections x 85x85 bins, NSF 5x[3x64] K=20"),
}


# dense-contraction FLOPs of the conditioner, per particle and per flow layer:
#   2 * (d*h + 2*h^2 + h*q*d)  with h = 64, q = 3K-1 = 59
def layer_flops(d: int, h: int = 64, hidden_layers: int = 3, q: int = 59) -> int:
    return 2 * (d * h + (hidden_layers - 1) * h * h + h * q * d)


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--per-gpu", type=int, default=None, help="particles per GPU (overrides the scaling mode's batch)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="with --scaling strong: the global particle count divided over the ranks (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-n1", action="store_true",
                    help="skip the extra `strong_n1` object of the default N = 1 line (C4's full 16 777 216 particles on one GPU)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--bwd-chunk", type=int, default=None, help="particles per flow-backward chunk (tuning)")
    ap.add_argument("--meas-samples", type=int, default=1_000_000, help="ground-truth samples behind the measurements")
    ap.add_argument("--fused-adamw", action="store_true",
                    help="torch.optim.AdamW(fused=True): one optimizer kernel instead of ~13 foreach launches (same update)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (mentflow_amd.graph.GraphedTrainStep): the launch-bound small-batch "
                         "regime, e.g. --per-gpu 25000; single GPU only")
    # test infrastructure only: run the ranks on the host-emulated kernel build (tests/emu) so that the launcher and the
    # multi-rank plumbing can be exercised in a GPU-less container.  The line it prints is marked as emulated.
    ap.add_argument("--test-emulator-lib", default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ===================================================================================================== launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Parent of a self-launched multi-rank run.  Touches neither torch nor HIP: it only starts N children (fresh
    interpreters, one per GPU), relays rank 0's stdout (the JSON line) and reaps them.  SIGTERM / SIGINT (a driver
    timeout, Ctrl-C) end exactly the children this process started — terminate, then kill — so that no rank is left
    holding a GPU in a barrier; the launcher then exits non-zero."""
    import signal
    n = args.gpus
    port = _free_port()
    children = []

    def reap(sig_name=None):
        for c in children:
            if c.poll() is None:
                c.terminate()
        deadline = time.time() + 20
        for c in children:
            try:
                c.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                c.kill()
                try:
                    c.wait(timeout=10)                # reap the killed rank: no zombie holding its GPU context open
                except subprocess.TimeoutExpired:
                    pass
        if sig_name:
            log(f"launcher: {sig_name} received, {len(children)} ranks stopped")

    def on_signal(signum, _frame):
        reap(signal.Signals(signum).name)
        sys.exit(128 + signum)

    old_handlers = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    rank0_out = None
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            # dmabuf IPC: the host driver of this pool does not support the legacy IPC mode, and without this setting RCCL's
            # (and torch's) cross-process buffer sharing fails with "hipIpcGetMemHandle: invalid argument".  The image
            # already exports it; it is (re)stated here so that a rank never starts without it (it is echoed in the line).
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", "4")
            out = subprocess.PIPE if r == 0 else sys.stderr
            # SIGTERM / SIGINT stay blocked from the fork until the child is in `children`: a signal that lands in between would
            # otherwise be handled by a reap() that does not know the rank just started.  A signal mask survives fork AND exec,
            # so the child restores the launcher's previous mask itself before it execs (the launcher is single-threaded and
            # has imported neither torch nor HIP: preexec_fn is safe here).
            blocked = signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM, signal.SIGINT})
            try:
                children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out,
                                                 preexec_fn=lambda: signal.pthread_sigmask(signal.SIG_SETMASK, blocked)))
            finally:
                signal.pthread_sigmask(signal.SIG_SETMASK, blocked)
        log(f"launcher: started {n} ranks (pids {[c.pid for c in children]}), rendezvous 127.0.0.1:{port}")
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = children[r]
                if r == 0 and rank0_out is None:
                    # rank 0 prints one line at the very end; communicate() also reaps it
                    try:
                        rank0_out, _ = c.communicate(timeout=0.5)
                    except subprocess.TimeoutExpired:
                        continue
                code = c.poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    log(f"launcher: rank {r} exited with code {code}")
                    rc = rc or code
            if rc:
                break
            time.sleep(0.2)
    finally:
        reap()                                       # no-op when every rank has exited; only the children started here
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    if rank0_out:
        sys.stdout.write(rank0_out.decode())
        sys.stdout.flush()
    return rc


# ===================================================================================================== worker
def host_cores() -> int:
    """Cores this process may actually use: the scheduler affinity, capped at the GPU box's per-GPU CPU share (16) —
    os.cpu_count() reports every core of the host and oversubscribes the OpenMP pool."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(prob, budget_s: float = 20.0, n_cpu: int = 25_000):
    """The oracle (eager dense restatement of the reference semantics) timed on this host: same problem, the
    reference's own batch size (experiments/rec_nd_1d/run_rings.sh:21), zero_grad + loss + backward + AdamW."""
    import torch
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
           "kind_note": "oracle restatement (eager dense PyTorch); the reference itself cannot run: its flow arithmetic "
                        "is zuko==1.3.1, absent from the reference tree and from this image",
           "sample": f"{steps} train steps of {n_cpu} particles (reference batch size) on the same workload, "
                     f"{el:.1f} s of CPU work; oracle = eager dense PyTorch restatement of the reference"}
    return out


# Fixed parity gate of the bench line, evaluated on the INITIAL parameters (the state every run starts from: the reference's
# default initialisation), before any warm-up step.  Achieved there: gradients 9e-5 of the largest entry, |dL| 2e-6.
GATE_GRAD = 3.0e-4             # parameter gradients vs the fp64 oracle, max error / largest entry
GATE_GRAD_VS_FP32 = 3.0e-4     # ... and vs the fp32 oracle (what the reference itself computes on a CPU)
GATE_H = 1.0e-4


def parity_gate(prob, n: int = 36864, gated: bool = True):
    """The only place (with cpu_baseline) where bench.py touches the oracle: one loss + backward of the SAME model on the
    GPU and in the oracle from one injected base draw (36 864 particles), so that the bench line carries the evidence that
    what is timed computes what the reference computes.  gated=True (initial parameters): fixed gates, a failure fails the
    run.  gated=False (after the timed training steps): the same numbers, reported as information — the gradient sums
    cancel more and more as the fit converges, so late-training errors are relative to a shrinking quantity."""
    import torch
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
    # the oracle in fp64 is the reference value; the same oracle in fp32 (what the reference itself would compute on the
    # CPU) is reported next to it
    Lo, Ho, Do, go = oracle_step(prob, z, torch.float64)
    L32, H32, D32, g32 = oracle_step(prob, z, torch.float32)
    gmax = float(go.abs().max())
    mu = float(model.penalty_parameter)
    res = {"particles": n, "reference": "oracle in fp64", "state": "initial parameters" if gated else "after the timed steps",
           "L_abs_err": abs(float(L) - float(Lo)), "H_abs_err": abs(float(H) - float(Ho)),
           "D_max_abs_err": float((torch.stack(D).detach().cpu().double() - torch.stack(Do)).abs().max()),
           "grad_max_err_over_max_grad": float((g.double() - go).abs().max() / gmax), "L": float(Lo),
           "fp32_oracle_vs_fp64": {"L_abs_err": abs(float(L32) - float(Lo)),
                                   "grad_max_err_over_max_grad": float((g32.double() - go).abs().max() / gmax)},
           "grad_max_err_vs_fp32_oracle": float((g.double() - g32.double()).abs().max() / gmax)}
    if gated:
        res["gates"] = {"grad": GATE_GRAD, "grad_vs_fp32_oracle": GATE_GRAD_VS_FP32, "H": GATE_H,
                        "L": 1e-4 + mu * 2e-6}
        res["ok"] = bool(res["L_abs_err"] < 1e-4 + mu * 2e-6 and res["H_abs_err"] < GATE_H
                         and res["grad_max_err_over_max_grad"] < GATE_GRAD
                         and res["grad_max_err_vs_fp32_oracle"] < GATE_GRAD_VS_FP32)
    log(("parity gate: " if gated else "parity after training (information only): ") + json.dumps(res))
    return res


def kde_issued_atomics(prob, x, cap: int = 8192):
    """LDS atomics the KDE forward kernels ISSUE per particle (all projections): window cells inside the grid whose
    fixed-point weight is non-zero (w >= 2^-50; kde.hip to_fix / kde2d_dead_cell), counted on a sample of the timed
    model's own particles with the kernels' arithmetic.  The window has 9 (1-D) / 69 (2-D) cells; the far ones carry
    weights below the quantum for most positions of the particle inside its cell and are skipped by the kernels."""
    import torch
    x = x[:cap].detach().double()
    d0 = prob.diagnostics[0][0]
    tiny = 2.0 ** -50
    total = 0.0
    with torch.no_grad():
        for t in prob.transforms:
            M = t.matrix.detach().to(x.device).double()
            u = x @ M.T
            if d0.ndim == 1:
                c = d0.coords.to(x.device).double()
                delta = float(c[1] - c[0])
                sig = float(d0.bandwidth_bins) * delta
                uu = u[:, d0.axis]
                kc = torch.round((uu - c[0]) / delta)
                cnt = torch.zeros_like(uu)
                for j in range(-4, 5):
                    k = kc + j
                    r = (uu - (c[0] + k * delta)) / sig
                    cnt += ((k >= 0) & (k < c.numel()) & (torch.exp(-0.5 * r * r) >= tiny)).double()
            else:
                cx, cy = d0.coords_x.to(x.device).double(), d0.coords_y.to(x.device).double()
                dx, dy = float(cx[1] - cx[0]), float(cy[1] - cy[0])
                bw = d0.bandwidth_bins
                sx, sy = float(bw[0]) * dx, float(bw[1]) * dy
                ux, uy = u[:, d0.axis[0]], u[:, d0.axis[1]]
                kx, ky = torch.round((ux - cx[0]) / dx), torch.round((uy - cy[0]) / dy)
                cnt = torch.zeros_like(ux)
                for i in range(-4, 5):
                    wx = torch.exp(-0.5 * ((ux - (cx[0] + (kx + i) * dx)) / sx) ** 2)
                    okx = (kx + i >= 0) & (kx + i < cx.numel())
                    for j in range(-4, 5):
                        a, b = abs(i), abs(j)
                        if a >= 1 and b >= 1 and (2 * a - 1) ** 2 + (2 * b - 1) ** 2 > 69:
                            continue                                     # kde2d_dead_cell: never visited
                        wy = torch.exp(-0.5 * ((uy - (cy[0] + (ky + j) * dy)) / sy) ** 2)
                        oky = (ky + j >= 0) & (ky + j < cy.numel())
                        cnt += (okx & oky & (wx * wy >= tiny)).double()
            total += float(cnt.mean())
    return total


def traffic_from_profile(dom: str, workload: str, per_gpu: int, fused_bwd: bool, act_level: int = 0):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json,
    written by tools/summarise_pmc.py).  A constant of that profiled run, NOT a measurement of this one: reported only
    when workload, per-GPU batch and backward variant are the ones that were profiled, with its provenance."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None, None
    meta = t.get("_meta", {})
    if meta.get("workload", "c4") != workload or int(meta.get("per_gpu", 2_097_152)) != per_gpu:
        return None, None
    if bool(meta.get("fused_bwd", True)) != fused_bwd or int(meta.get("act_level", 0)) != act_level or dom not in t:
        return None, None
    src = {"file": "profiles/traffic.json", "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, builder run "
           "(FETCH_SIZE x2 on gfx950); constant of that run, not measured by this one", **meta}
    return t[dom], src


def pipe_util_from_profile(name: str, workload: str, per_gpu: int, act_level: int):
    """Fraction of the SIMD cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 x 1024
    SIMDs), from the committed rocprofv3 PMC pass — the dense-equivalent `frac` counts the FLOPs of the masked-out blocks the
    kernels skip, this is what the pipe really did.  Like `traffic` a constant of the profiled run, reported with provenance
    and only for the workload / batch / hand-off level that was profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None, None
    meta = t.get("_meta", {})
    pu = t.get("_mfma_pipe_busy", {})
    if (meta.get("workload", "c4") != workload or int(meta.get("per_gpu", 2_097_152)) != per_gpu
            or int(meta.get("act_level", 0)) != act_level or name not in pu):
        return None, None
    return pu[name], {"file": "profiles/traffic.json", "how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, builder run: "
                      "busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); constant of that run", **meta}


def run_worker(args) -> int:
    import torch

    sys.path.insert(0, ROOT)
    import mentflow_amd as mf                                   # noqa: F401
    from mentflow_amd import _lib
    from mentflow_amd import dist as mfdist
    from mentflow_amd.harness import build_problem

    emulated = args.test_emulator_lib is not None
    if emulated:
        _lib.use_library(args.test_emulator_lib)
    want_world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (not emulated and want_world > 1 and os.environ.get("MENTFLOW_SHARE_GPU") != "1"
            and torch.cuda.device_count() <= local):
        raise SystemExit(f"rank {local}: --gpus {want_world} needs {want_world} visible GPUs, found "
                         f"{torch.cuda.device_count()} (MENTFLOW_SHARE_GPU=1 rehearses all ranks on cuda:0 over gloo)")
    device = mfdist.init_from_env(backend="gloo" if emulated else None)
    world = mfdist.world_size()
    rank = mfdist.rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if device.type != "cuda" and not emulated:
        raise SystemExit("bench.py needs an MI355X: mentflow_amd has no CPU path")
    if emulated:
        device = torch.device("cpu")

    w = dict(WORKLOADS[args.workload])
    weak_per_gpu = w.pop("per_gpu")
    strong_global = w.pop("global_batch")
    desc = w.pop("desc")
    if args.per_gpu:
        per_gpu, global_batch = args.per_gpu, args.per_gpu * world
    elif args.scaling == "strong":
        global_batch = args.global_batch or strong_global
        if global_batch % world:
            raise SystemExit(f"--scaling strong: the global batch {global_batch} is not divisible by {world} ranks")
        per_gpu = global_batch // world
    else:
        per_gpu, global_batch = weak_per_gpu, weak_per_gpu * world
    prob = build_problem(device=device, penalty_parameter=500.0, meas_samples=args.meas_samples, **w)  # same seed: same weights
    model = prob.model
    gspec = model.generator.spec() if hasattr(model.generator, "spec") else None
    act_level = gspec.resolve_act_level(per_gpu, device) if gspec is not None else 0
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0, capturable=bool(args.graph),
                            **({"fused": True} if args.fused_adamw else {}))                             # experiments/setup.py:166-170
    torch.manual_seed(1234 + rank)                                           # every rank draws its own particles
    if args.graph and (world > 1 or emulated):
        raise SystemExit("--graph is a single-GPU mode")
    if args.bwd_chunk:
        model.generator.spec().bwd_chunk = args.bwd_chunk

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    if args.graph:
        gstep = mf.graph.GraphedTrainStep(model, opt, global_batch)

        def step():
            return gstep.step()[0]
    else:
        def step():
            opt.zero_grad()
            L, H, D = model.loss(global_batch)
            L.backward()
            opt.step()
            return L

    # evidence that the collective really spans the ranks: all-reduce of ones over the production backend
    ranks_seen = 1
    backend = "none"
    if world > 1:
        ones = torch.ones(1, dtype=torch.float32, device=device)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(round(float(ones[0])))
        backend = torch.distributed.get_backend()

    # parity gate on the INITIAL parameters (before any optimizer step), fixed gates: a failure fails the run
    parity0 = None
    if not args.no_cpu_baseline and world == 1 and not emulated and rank == 0:
        parity0 = parity_gate(prob, gated=True)

    for _ in range(args.warmup):
        step()
    region_s = []
    rank_s = []
    _lib.prof_enable(True)
    for rep in range(max(1, args.repeats)):
        sync()
        mfdist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            L = step()
        sync()
        mfdist.barrier()
        sync()
        el = time.perf_counter() - t0
        per_rank = [el]
        if world > 1:
            # every rank's own clock around the same region: the line reports the MAX (the contract) and keeps min / max
            # / all of them, so that a straggler rank is visible
            t = torch.zeros(world, dtype=torch.float64, device=device)
            t[rank] = el
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
            per_rank = [float(v) for v in t.cpu()]
            el = max(per_rank)
        region_s.append(el)
        rank_s.append(per_rank)
    prof = _lib.prof_report()
    _lib.prof_enable(False)
    final_loss = float(L.detach())
    total_steps = args.steps * len(region_s)

    # The metric's own full-size configuration in the same record: C4's GLOBAL batch (16 777 216 particles, the N = 1 point of
    # the strong-scaling series; experiments/rec_nd_1d/run_gmm.sh:32-41 scaled as BASELINE C4) on this one GPU, a few steps
    # after the timed regions.  `value` stays the weak point above; this object is information.
    strong_n1 = None
    if (world == 1 and rank == 0 and not emulated and not args.graph and not args.no_strong_n1 and args.workload == "c4"
            and args.scaling == "weak" and not args.per_gpu):
        try:
            big = strong_global
            big_level = gspec.resolve_act_level(big, device) if gspec is not None else 0
            sn_steps, sn_warm = 3, 1

            def big_step():
                opt.zero_grad()
                Lb, _, _ = model.loss(big)
                Lb.backward()
                opt.step()
                return Lb

            for _ in range(sn_warm):
                big_step()
            _lib.prof_enable(True)
            sync()
            t0 = time.perf_counter()
            for _ in range(sn_steps):
                Lb = big_step()
            sync()
            el_b = time.perf_counter() - t0
            prof_b = _lib.prof_report()
            _lib.prof_enable(False)
            strong_n1 = {"workload": "c4 at its global batch on ONE GPU (the N = 1 point of --scaling strong)", "global_batch": big,
                         "steps": sn_steps, "warmup": sn_warm, "ms_per_step": el_b / sn_steps * 1e3,
                         "value": big * sn_steps / el_b, "unit": "particle-samples/s", "activation_handoff_level": big_level,
                         "kernel_ms_per_step": {k: v[0] / sn_steps for k, v in prof_b.items() if v[1] > 0},
                         "peak_memory_GB": torch.cuda.max_memory_allocated(device) / 1e9, "final_loss": float(Lb.detach())}
            del Lb
            torch.cuda.empty_cache()
        except Exception as exc:                                  # information only: never fail the headline run for it
            _lib.prof_enable(False)
            strong_n1 = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        log("strong_n1: " + json.dumps(strong_n1))

    rc = 0
    if rank == 0:
        order = sorted(range(len(region_s)), key=lambda i: region_s[i])
        med = order[len(order) // 2]
        elapsed = region_s[med]                                             # median region
        d = w["ndim"]
        maf = w.get("gen_name") == "maf"
        lf = layer_flops(d, q=2 if maf else 59)                             # MAF: shift + scale per feature
        T = w["transforms"]
        P = w["num"]
        value = global_batch * args.steps / elapsed
        out = {
            "metric": "particle-samples/sec per MENT-Flow train step (6D, 100 proj)" if args.workload == "c4"
                      else f"particle-samples/sec per MENT-Flow train step ({args.workload})",
            "value": value, "unit": "particle-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not emulated else "synthetic, HOST-EMULATED kernels (test run)",
            "backend": backend, "ranks_seen": ranks_seen,
            "timed_regions": {"count": len(region_s), "steps_each": args.steps, "reported": "median",
                              "ms_per_step": [s / args.steps * 1e3 for s in region_s],
                              "min_ms_per_step": min(region_s) / args.steps * 1e3,
                              "max_ms_per_step": max(region_s) / args.steps * 1e3},
            "per_rank_ms_per_step": {"region": "the reported (median) region", "ranks": [v / args.steps * 1e3 for v in rank_s[med]],
                                     "min": min(rank_s[med]) / args.steps * 1e3, "max": max(rank_s[med]) / args.steps * 1e3},
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
            "config": {"workload": f"{args.workload}: {desc}", "global_batch": global_batch, "per_gpu_batch": per_gpu,
                       "parallelism": f"dp{world} (particle batch sharded; 2 all-reduces/step)",
                       "step": "zero_grad + MENTFlow.loss + backward + AdamW.step" + (" (hipGraph replay)" if args.graph else ""),
                       "activation_handoff": {"level": act_level, "meaning": "0 = the backward recomputes the conditioner; 1 = the "
                                              "forward hands its hidden tiles to the backward through HBM; 2 = also the "
                                              "conditioner outputs (mf_flow_rqs_layer_fwd_save / _bwd_saved)",
                                              "bytes_per_particle_and_layer": (4 * _lib.get_lib().mf_flow_rqs_act_floats(
                                                  32, gspec.d, gspec.L, gspec.bins, act_level) // 32) if act_level else 0},
                       "final_loss": final_loss},
        }
        kernels = {k: v for k, v in prof.items() if v[1] > 0}
        if kernels:
            # the fused backward kernel (no outer_accum launches) does backward-data AND the parameter gradients: 2 F_layer
            fused_bwd = "outer_accum" not in kernels
            issued = {}                                  # KDE forward: issued LDS atomics per particle, counted once

            def kernel_roofline(name):
                """Roofline entry of one timed kernel: algorithmic work per launch / HIP-event average launch duration."""
                ms, cnt = kernels[name]
                flow_k = name in ("flow_layer_fwd", "flow_layer_bwd", "outer_accum")
                per_launch_particles = per_gpu * (T if flow_k else 1) / (cnt / total_steps)
                avg_s = ms / cnt * 1e-3
                alg_flops = {"flow_layer_fwd": lf, "flow_layer_bwd": 2 * lf if fused_bwd else lf, "outer_accum": lf}.get(name)
                roof = {"kernel": name + (" (fused: backward-data + parameter gradients)" if name == "flow_layer_bwd" and fused_bwd else "")}
                if alg_flops is not None:
                    roof.update(bound="mfma", unit="TFLOP/s", peak=PEAK_MFMA_F32,
                                achieved=alg_flops * per_launch_particles / avg_s / 1e12,
                                algorithmic_per_launch=f"{alg_flops} FLOP/particle x {int(per_launch_particles)} particles")
                elif name.endswith("_fwd"):
                    # KDE forward: every particle adds fixed-point weights to the (2R+1) [1-D] or (2R+1)^2 - 12 [2-D] window
                    # cells of each projection with 64-bit LDS atomics — those with a non-zero weight; the ceiling is the
                    # measured ds_add_u64 issue rate of the chip, not HBM
                    visited = P * (9 if name == "kde1d_fwd" else 69)
                    if name not in issued:
                        with torch.no_grad():
                            issued[name] = kde_issued_atomics(prob, model.generator.sample(8192))
                    ops_pp = issued[name]
                    peak = LDS_ATOMIC_U64_PER_CLK_CU * NUM_CU * CLOCK_GHZ            # G lane-atomics / s
                    roof.update(bound="lds_atomic", unit="G ds_add_u64/s", peak=peak,
                                peak_source="builder-measured: tools/ubench_lds_atomics2.hip, profiles/r02_ubench_lds_atomics2.txt "
                                            "(3.9 ds_add_u64 per clock and CU x 256 CUs x 2.4 GHz); not a guide number",
                                achieved=ops_pp * per_launch_particles / avg_s / 1e9,
                                algorithmic_per_launch=f"{ops_pp:.1f} issued LDS atomics/particle (of {visited} window cells visited; "
                                                       f"counted on 8192 of the model's particles, weights >= 2^-50) x "
                                                       f"{int(per_launch_particles)} particles",
                                hbm_GBps=4 * d * per_launch_particles / avg_s / 1e9)
                else:
                    # KDE backward: per particle and projection a 9-term (9 x 9 in 2-D) window of exp / gather / FMA work and
                    # 8 d bytes of row traffic per particle (76 GB/s at C4: two orders under HBM).  What bounds it is vector
                    # issue: the loop body's instruction mix (KDE_BWD_MIX, from the ISA) against the SIMDs' issue capacity.
                    nv, nt, nl = KDE_BWD_MIX[name]
                    simd_cycles = VALU_CYCLES * nv + TRANS_CYCLES * nt            # per wave (64 lanes) and projection
                    pp = P * per_launch_particles                                 # particle-projections per launch
                    peak = SIMDS * CLOCK_GHZ * 64 / simd_cycles                   # G particle-projections / s
                    roof.update(bound="valu_issue", unit="G particle-projections/s", peak=peak, achieved=pp / avg_s / 1e9,
                                peak_source=f"{nv} full-rate VALU x {VALU_CYCLES} cycles + {nt} transcendental x {TRANS_CYCLES} "
                                            f"cycles per wave64 and projection (inner loop of the gfx950 ISA, tools/isa_loop_mix.py, "
                                            f"profiles/r04_kde_bwd_instruction_mix.txt) on {SIMDS} SIMDs at {CLOCK_GHZ} GHz; "
                                            f"{nl} LDS reads per lane ride along (not booked)",
                                algorithmic_per_launch=f"{P} projections x {int(per_launch_particles)} particles",
                                hbm_GBps=8 * d * per_launch_particles / avg_s / 1e9)
                roof["frac"] = roof["achieved"] / roof["peak"]
                roof["avg_launch_ms"] = ms / cnt
                roof["launches"] = cnt
                return roof

            # dominant kernel by summed HIP-event time inside the timed regions
            dom = max(kernels, key=lambda k: kernels[k][0])
            roof = kernel_roofline(dom)
            roof["traffic"], roof["traffic_source"] = traffic_from_profile(dom, args.workload, per_gpu, fused_bwd, act_level)
            out["roofline"] = roof
            # the same accounting for every timed kernel (the dominant one is `roofline`)
            out["kernel_rooflines"] = {k: {q: v for q, v in kernel_roofline(k).items()
                                           if q in ("bound", "unit", "peak", "achieved", "frac", "avg_launch_ms",
                                                    "algorithmic_per_launch", "peak_source", "hbm_GBps")}
                                       for k in kernels}
            # `frac` of the MFMA-bound kernels is DENSE-EQUIVALENT (FLOPs as the reference computes them): the kernels skip
            # the MFMAs of all-zero mask blocks, so a frac of 0.9 does not mean the pipe is 90 % busy.  What the pipe did:
            for k, r in list(out["kernel_rooflines"].items()) + [(dom, roof)]:
                if r.get("bound") == "mfma":
                    r["mfma_pipe_busy"], r["mfma_pipe_busy_source"] = pipe_util_from_profile(k, args.workload, per_gpu, act_level)
            out["step_mfma_frac"] = 3 * T * lf * (value / world) / (PEAK_MFMA_F32 * 1e12)
            out["kernel_ms_per_step"] = {k: v[0] / total_steps for k, v in kernels.items()}
        if strong_n1 is not None:
            out["strong_n1"] = strong_n1
        log("gpu leg: " + json.dumps({k: out.get(k) for k in ("value", "ms_per_step", "timed_regions", "roofline",
                                                              "kernel_ms_per_step")}))
        if world > 1 and ranks_seen != world:
            log(f"ERROR: the all-reduce saw {ranks_seen} ranks, expected {world}")
            rc = 3
        if not args.no_cpu_baseline and world == 1 and not emulated:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_budget)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            out["cpu_baseline"]["parity"] = parity0
            out["cpu_baseline"]["parity_after_training"] = parity_gate(prob, gated=False)
            if not parity0["ok"]:
                # a kernel that does not compute what the reference computes has no throughput worth reporting
                log("ERROR: parity gate FAILED on the initial parameters: " + json.dumps(parity0))
                out["value"], out["parity_failed"], rc = None, True, 4
        print(json.dumps(out), flush=True)
    mfdist.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return rc


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)          # before any torch / HIP call in this process
    return run_worker(args)


if __name__ == "__main__":
    sys.exit(main())
