#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -2 $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02i/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), "ms/step %.3f" % j["ms_per_step"], {k: round(v,3) for k,v in j.get("kernel_ms_per_step",{}).items()}, "frac %.4f" % j["roofline"]["frac"])
    if "cpu_baseline" in j: print(json.dumps(j["cpu_baseline"]["parity"]))
PY
