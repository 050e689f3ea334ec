#!/bin/bash
# Development tool, runs ON THE GPU BOX: hand-off parity tests, then one bench line per activation hand-off level
# (MENTFLOW_ACT_LEVEL = 0 recompute, 1 hidden tiles, 2 + conditioner outputs).  Usage: tools/ab_levels.sh [levels...]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/levels; mkdir -p $OUT
BARGS=${AB_BENCH_ARGS:---steps 10 --warmup 3 --repeats 3 --no-cpu-baseline}
if [ "${AB_TESTS:-1}" = "1" ]; then
  timeout -k 10 900 python -m pytest tests/test_activation_handoff.py tests/test_fused_backward.py -m gpu -x -q > $OUT/pytest.txt 2>&1 \
    || { echo "== PARITY TESTS FAILED"; tail -30 $OUT/pytest.txt; exit 1; }
  echo "== $(tail -1 $OUT/pytest.txt)"
fi
for lv in ${@:-0 1 2}; do
  MENTFLOW_ACT_LEVEL=$lv timeout -k 10 300 python bench.py $BARGS > $OUT/level$lv.json 2> $OUT/level$lv.err || { echo "== level $lv: bench failed"; tail -5 $OUT/level$lv.err; exit 1; }
  python - $lv <<'PY'
import json, sys
lv = sys.argv[1]
j = json.loads(open(f"gpurun_out/levels/level{lv}.json").read().strip().splitlines()[-1])
print(f"== level {lv}: ms/step {j['ms_per_step']:.3f}", {k: round(x, 3) for k, x in j['kernel_ms_per_step'].items()}, "frac %.4f" % j['roofline']['frac'])
PY
done
