#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/kernel_census.py > $OUT/census_default.txt 2>&1 || tail -20 $OUT/census_default.txt
python tools/kernel_census.py --fused > $OUT/census_fused.txt 2>&1 || tail -20 $OUT/census_fused.txt
head -60 $OUT/census_default.txt
echo ===== fused
head -25 $OUT/census_fused.txt
python bench.py --workload c3 --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --graph --fused-adamw > $OUT/bench_c3_25k_graph_fused.json 2> $OUT/bench_c3_25k_graph_fused.err
python bench.py --workload c3 --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --fused-adamw > $OUT/bench_c3_25k_eager_fused.json 2> $OUT/bench_c3_25k_eager_fused.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02f/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), "ms/step %.3f" % j["ms_per_step"])
PY
