#!/bin/bash
# Development tool, runs ON THE GPU BOX: A/B prebuilt library variants (variants/<name>.so, built locally with hipcc).
# For each name: flow parity tests against the oracle, then one C4 bench line.  Usage: tools/ab_variants.sh base new ...
# env: AB_TESTS=0 skips the parity tests, AB_BENCH_ARGS overrides the bench arguments.
set -e
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab; mkdir -p $OUT
BARGS=${AB_BENCH_ARGS:---steps 10 --warmup 3 --repeats 3 --no-cpu-baseline}
for v in "$@"; do
  cp variants/$v.so mentflow_amd/csrc/libmentflow_hip.so
  if [ "${AB_TESTS:-1}" = "1" ]; then
    if ! timeout -k 10 600 python -m pytest tests/test_fused_backward.py tests/test_flow_kernels.py -m gpu -x -q > $OUT/$v.pytest.txt 2>&1; then
      echo "== $v: PARITY TESTS FAILED"; tail -30 $OUT/$v.pytest.txt; continue
    fi
    echo "== $v: $(tail -1 $OUT/$v.pytest.txt)"
  fi
  timeout -k 10 300 python bench.py $BARGS > $OUT/$v.json 2> $OUT/$v.err || { echo "== $v: bench failed"; tail -5 $OUT/$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads(open(f"gpurun_out/ab/{v}.json").read().strip().splitlines()[-1])
print(f"== {v}: ms/step {j['ms_per_step']:.3f}", {k: round(x, 3) for k, x in j['kernel_ms_per_step'].items()}, "frac %.4f" % j['roofline']['frac'],
      "parity", {k: (f"{x:.2e}" if isinstance(x, float) else x) for k, x in (j.get('cpu_baseline') or {}).get('parity', {}).items()} if j.get('cpu_baseline') else "")
PY
done
