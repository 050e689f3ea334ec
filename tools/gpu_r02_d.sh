#!/bin/bash
# Runs ON THE GPU BOX: full GPU suite after the always-fused default + graphed Trainer; small-batch benches eager / graph.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python tools/fb_diag.py > $OUT/fb_diag.txt 2>&1 && tail -18 $OUT/fb_diag.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c4_25k_eager.json 2> $OUT/bench_c4_25k_eager.err && echo "25k eager done" &&
python bench.py --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --graph > $OUT/bench_c4_25k_graph.json 2> $OUT/bench_c4_25k_graph.err && echo "25k graph done" &&
python bench.py --workload c3 --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --graph > $OUT/bench_c3_25k_graph.json 2> $OUT/bench_c3_25k_graph.err && echo "c3 25k graph done" &&
python bench.py --workload c1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c1_eager.json 2> $OUT/bench_c1_eager.err &&
python bench.py --workload c1 --steps 200 --warmup 20 --no-cpu-baseline --graph > $OUT/bench_c1_graph.json 2> $OUT/bench_c1_graph.err && echo "c1 done" &&
python examples/train_rec_nd_1d.py --epochs 4 --iters 150 > $OUT/example_rings.txt 2>&1 && tail -8 $OUT/example_rings.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02d/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e ms/step %.3f" % (j["value"], j["ms_per_step"]), {k: round(v,3) for k,v in j.get("kernel_ms_per_step",{}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
