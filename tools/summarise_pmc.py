#!/usr/bin/env python
"""Turns the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/profiles_<tag>/) into the tracked summaries under
profiles/: kernel stats, per-kernel HBM traffic per launch (FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md
§HBM prescribes for wide coalesced reads on gfx950; WRITE_SIZE as reported), SQ utilisation, and traffic.json, which
bench.py reads to fill roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

KEY = {"rqs_layer_fwd_kernel": "flow_layer_fwd", "rqs_layer_bwd_kernel": "flow_layer_bwd",
       "rqs_layer_bwd_fused_kernel": "flow_layer_bwd", "outer_accum_kernel": "outer_accum",
       "proj_kde1d_fwd_kernel": "kde1d_fwd", "proj_kde1d_bwd_kernel": "kde1d_bwd", "proj_kde2d_fwd_kernel": "kde2d_fwd",
       "proj_kde2d_bwd_kernel": "kde2d_bwd"}


def short(name):
    for k, v in KEY.items():
        if k in name:
            return v
    return None


def one(pattern):
    # gpurun merges every run's output into the same scratch tree: take the NEWEST match, not the first
    files = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


stats = one("stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f in ("stats/bench.json", "bench_default.json", "bench_strong_n1.json", "bench_c5.json", "bench_2ranks_shared.json",
          "bench_c1.json", "bench_c2.json", "bench_c3.json", "bench_c3_25k_eager.json", "bench_c3_25k_graph_fused.json",
          "fused_bwd_cycles.txt", "fused_bwd_ablation_no_barrier.txt", "fused_bwd_ablation_no_dw.txt",
          "fused_bwd_ablation_no_spline.txt", "bench_level0.json", "bench_level1.json", "bench_level2.json",
          "c5_stats/bench.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_" + f.replace("/", "_")))


def per_kernel(csv_path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(csv_path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            if k:
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


traffic = {}
fetch, write = one("fetch/**/*counter_collection.csv"), one("write/**/*counter_collection.csv")
rows = []
if fetch and write:
    F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    for k in sorted(set(F) | set(W)):
        f_kb = F.get(k, (0, 0))[0]
        w_kb = W.get(k, (0, 0))[0]
        hbm = (2.0 * f_kb + w_kb) * 1024.0            # FETCH_SIZE counts 64 B per 128-B request on gfx950: doubled
        traffic[k] = hbm
        rows.append((k, F.get(k, (0, 0))[1], f_kb, w_kb, hbm))
    with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KB_per_launch(raw),WRITE_SIZE_KB_per_launch,HBM_bytes_per_launch(2*fetch+write)\n")
        for r in rows:
            fh.write("%s,%d,%.1f,%.1f,%.0f\n" % r)
    # provenance: bench.py reports these numbers only for the workload / batch / backward variant that was profiled
    meta = {"workload": "c4", "per_gpu": 2097152, "fused_bwd": True, "tag": tag,
            "command": "python3 bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline"}
    try:      # the activation hand-off level the profiled bench ran at (bench.py echoes it in its line)
        bj = json.loads(open(os.path.join(src, "stats", "bench.json")).read().strip().splitlines()[-1])
        meta["act_level"] = int(bj["config"]["activation_handoff"]["level"])
    except Exception:
        meta["act_level"] = 0
    try:
        meta["commit"] = open(os.path.join(src, "commit.txt")).read().strip()
    except OSError:
        pass
    traffic["_meta"] = meta
    # matrix-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) against the SIMD cycles of the
    # same dispatches, GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs (MI355X_MICROARCH.md: the counter is the sum over the 8 XCDs)
    sqf = one("sq/**/*counter_collection.csv")
    if sqf:
        busy, act = collections.defaultdict(float), collections.defaultdict(float)
        for r in csv.DictReader(open(sqf)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[k] += float(r["Counter_Value"])
            if k and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                act[k] += float(r["Counter_Value"])
        traffic["_mfma_pipe_busy"] = {k: busy[k] / (act[k] / 8.0 * 1024.0) for k in busy if act.get(k, 0) > 0 and busy[k] > 0}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

c5stats = one("c5_stats/**/*kernel_stats.csv")
if c5stats:
    shutil.copy(c5stats, os.path.join(dst, f"{tag}_c5_kernel_stats.csv"))
for sub, name in (("sq", f"{tag}_sq_counters.csv"), ("c5_sq", f"{tag}_c5_sq_counters.csv")):
  sq = one(sub + "/**/*counter_collection.csv")
  if sq:
      agg = collections.defaultdict(lambda: collections.defaultdict(float))
      for r in csv.DictReader(open(sq)):
          k = short(r["Kernel_Name"])
          if k:
              agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
      names = sorted({n for v in agg.values() for n in v})
      with open(os.path.join(dst, name), "w") as fh:
          fh.write("kernel," + ",".join(names) + "\n")
          for k, v in agg.items():
              fh.write(k + "," + ",".join("%.4g" % v[n] for n in names) + "\n")
print("wrote", sorted(os.listdir(dst)))
