#!/bin/bash
# Runs ON THE GPU BOX: fused-backward parity tests + one C4 bench line (kernel iteration loop)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fused_backward.py tests/test_flow_kernels.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -1 $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python -c "
import json; j=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); print('c4 ms/step %.3f' % j['ms_per_step'], {k: round(v,3) for k,v in j['kernel_ms_per_step'].items()}, 'frac %.4f' % j['roofline']['frac'])"
