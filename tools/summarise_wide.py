#!/usr/bin/env python
"""gpurun_out/wide_<tag>/ (tools/collect_wide_profile.sh) -> profiles/<tag>_wide128_kernels.csv and <tag>_wide128_bench.json:
per kernel of the wide-conditioner step the calls, average duration, HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, as tools/summarise_pmc.py) and the matrix-pipe busy fraction
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"wide_{tag}")
NAMES = {"wide_fwd_kernel": "wide_fwd", "wide_bwd_kernel": "wide_bwd", "wide_outer_accum_kernel": "wide_outer_accum",
         "proj_kde1d_fwd": "kde1d_fwd", "proj_kde1d_bwd": "kde1d_bwd", "grad_reduce": "grad_reduce"}


def one(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return files[-1]


def short(name):
    for k, v in NAMES.items():
        if k in name:
            return v
    return None


def per_kernel(pattern, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(one(pattern))):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            if k:
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


stats = {}
for r in csv.DictReader(open(one("stats/**/*kernel_stats.csv"))):
    k = short(r["Name"])
    if k:
        stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"]))
F, W = per_kernel("fetch/**/*counter_collection.csv", "FETCH_SIZE"), per_kernel("write/**/*counter_collection.csv", "WRITE_SIZE")
M, G = (per_kernel("sq/**/*counter_collection.csv", "SQ_VALU_MFMA_BUSY_CYCLES"),
        per_kernel("sq/**/*counter_collection.csv", "GRBM_GUI_ACTIVE"))
rows = ["kernel,calls,avg_ms,percent,HBM_bytes_per_launch(2*FETCH+WRITE),mfma_pipe_busy(SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE/8*1024))"]
for k, (calls, ms, pc) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    hbm = (2 * F.get(k, 0.0) + W.get(k, 0.0)) * 1024
    busy = M[k] / (G[k] / 8 * 1024) if k in M and G.get(k) else float("nan")
    rows.append(f"{k},{calls},{ms:.4f},{pc:.2f},{hbm:.0f},{busy:.3f}")
open(os.path.join(root, "profiles", f"{tag}_wide128_kernels.csv"), "w").write("\n".join(rows) + "\n")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(root, "profiles", f"{tag}_wide128_bench.json"))
print("\n".join(rows))
