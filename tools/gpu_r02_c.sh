#!/bin/bash
# Runs ON THE GPU BOX: chain micro-benchmark, full GPU test-suite, benches (C4 default, C5, C1, C2, C3, 25 k reference batch).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
./tools/bin/ubench_chain > $OUT/ubench_chain.txt 2>&1 || echo "ubench failed"
cat $OUT/ubench_chain.txt
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python bench.py --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err && echo "c4 done" &&
python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err && echo "c5 done" &&
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err && echo "c3 done" &&
python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err && echo "c2 done" &&
python bench.py --workload c1 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_c1.json 2> $OUT/bench_c1.err && echo "c1 done" &&
python bench.py --per-gpu 25000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_c4_25k.json 2> $OUT/bench_c4_25k.err && echo "25k done" &&
MENTFLOW_BWD_FUSED=1 python bench.py --per-gpu 25000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_c4_25k_fused.json 2> $OUT/bench_c4_25k_fused.err && echo "25k fused done"
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02c/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e ms/step %.3f" % (j["value"], j["ms_per_step"]), {k: round(v,3) for k,v in j.get("kernel_ms_per_step",{}).items()}, "roof", round(j["roofline"]["frac"],3), j["roofline"]["kernel"][:20])
    except Exception as e:
        print(f, "ERR", e)
PY
