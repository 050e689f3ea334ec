#!/bin/bash
# Development tool, runs ON THE GPU BOX: dynamic instruction mix of the flow kernels for prebuilt library variants
# (variants/<name>.so).  One rocprofv3 PMC pass per variant (kernel-trace only), summed per kernel over a short C4 bench.
set -e
cd "$GRAFT_REPO_ROOT"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT
CTRS=${PMC_COUNTERS:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY}
for v in "$@"; do
  cp variants/$v.so mentflow_amd/csrc/libmentflow_hip.so
  rm -rf $OUT/$v; mkdir -p $OUT/$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/$v -- python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $OUT/$v/bench.json 2> $OUT/$v/err.txt) || { echo "$v: rocprofv3 failed"; tail -5 $OUT/$v/err.txt; continue; }
  python3 - "$v" "$OUT/$v" <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
if not f: print(v, "no counter csv"); sys.exit(0)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[-1])):
    k = r["Kernel_Name"]
    for key in ("rqs_layer_bwd_fused", "rqs_layer_fwd"):
        if key in k:
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for key in tot:
    n = max(cnt[(key, c)] for c in tot[key])
    print(f"{v:12s} {key:22s} per launch:", "  ".join(f"{c.replace('SQ_', '')}={tot[key][c] / n:.4g}" for c in sorted(tot[key])))
PY
done
