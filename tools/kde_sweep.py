"""Development tool (runs on the GPU box): times the projection + KDE kernels at the C4 / C5 shapes for one setting of
the MENTFLOW_KDE* tuning variables (read once per process), or — with --sweep — spawns itself over a grid of settings.
    python tools/kde_sweep.py --sweep 1d|2d  > gpurun_out/kde_sweep.txt"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(kind: str, n: int):
    import torch
    from mentflow_amd import ops
    from mentflow_amd.harness import build_problem
    dev = torch.device("cuda", 0)
    optics = "nd_1d" if kind == "1d" else "nd_2d_random"
    bins = 64 if kind == "1d" else 85
    prob = build_problem(ndim=6, num=100, bins=bins, xmax=3.5, seed=0, transforms=1, prior_scale=3.0, device=dev,
                         dist_name="gaussian_mixture", optics=optics, meas_samples=20000)
    diag = prob.diagnostics[0][0]
    torch.manual_seed(0)
    x = (torch.randn(n, 6, device=dev) * 1.2).requires_grad_(True)
    if kind == "1d":
        V = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
        f = lambda: ops.ProjKde1dFn.apply(x, V, diag.coords, float(diag.bandwidth), 4)
    else:
        V0 = torch.stack([t.matrix[0] for t in prob.transforms]).contiguous()
        V1 = torch.stack([t.matrix[2] for t in prob.transforms]).contiguous()
        f = lambda: ops.ProjKde2dFn.apply(x, V0, V1, diag.coords_x, diag.coords_y, float(diag.bandwidth_x),
                                          float(diag.bandwidth_y), 4, 4)
    S = f()
    g = torch.randn_like(S)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 5
    for it in range(reps + 1):
        x.grad = None
        ev[0].record(); S = f(); ev[1].record(); S.backward(g); ev[2].record()
        torch.cuda.synchronize()
        if it:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / reps, tb / reps, float(S.double().sum()), float(x.grad.double().abs().sum())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="1d")
    ap.add_argument("--n", type=int, default=2_097_152)
    ap.add_argument("--sweep", default=None)
    a = ap.parse_args()
    if a.sweep is None:
        tf, tb, cs, cg = one(a.kind, a.n)
        print(json.dumps({"fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3), "sumS": cs, "sum|gx|": cg}))
        sys.exit(0)
    if a.sweep == "1d":
        grid = [dict(MENTFLOW_KDE1D_BLOCK=b, MENTFLOW_KDE1D_LDS=l, MENTFLOW_KDE1D_WAVES=w)
                for b in (256, 512, 1024) for l in (53000, 79872, 159000) for w in (2, 4, 8)]
        grid += [dict(MENTFLOW_KDE1D_BWD_BLOCK=b, MENTFLOW_KDE1D_BWD_LDS=l) for b in (256, 512, 1024) for l in (27000, 53000, 98304)]
    else:
        grid = [dict(MENTFLOW_KDE2D_BLOCK=b, MENTFLOW_KDE2D_FWD_LDS=l, MENTFLOW_KDE2D_WAVES=w)
                for b in (256, 512, 1024) for l in (59000, 118000) for w in (4, 8, 16)]
        grid += [dict(MENTFLOW_KDE2D_BWD_BLOCK=b, MENTFLOW_KDE2D_BWD_NPT=p, MENTFLOW_KDE2D_BWD_LDS=l)
                 for b in (256, 512, 1024) for p in (1, 2, 4) for l in (30000, 59000, 118000)]
    for cfg in grid:
        env = dict(os.environ)
        env.update({k: str(v) for k, v in cfg.items()})
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--kind", a.sweep, "--n", str(a.n)], env=env,
                           capture_output=True, text=True)
        out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + r.stderr.strip()[-300:])
        print(json.dumps(cfg), out, flush=True)
