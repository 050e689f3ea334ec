#!/bin/bash
# Runs ON THE GPU BOX: C4 train-step throughput on one GPU against the per-GPU particle batch (the per-rank batches of the
# strong-scaling series are 16 777 216 / N: 8 M, 4 M, 2 M for N = 2, 4, 8).  One line per batch into gpurun_out/batch_sweep.txt.
set -e
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/batch_sweep.txt; : > $OUT
for n in 25000 100000 400000 1048576 2097152 4194304 8388608 16777216; do
  steps=10; [ $n -ge 8388608 ] && steps=5
  python3 bench.py --per-gpu $n --steps $steps --warmup 3 --repeats 3 --no-cpu-baseline 2> gpurun_out/batch_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('%9d particles  %9.3f ms/step  %.3e particle-samples/s   flow fwd %.3f  bwd %.3f  kde fwd %.3f  bwd %.3f  other %.3f' % ($n, d['ms_per_step'], d['value'], k['flow_layer_fwd'], k['flow_layer_bwd'], k['kde1d_fwd'], k['kde1d_bwd'], d['ms_per_step']-sum(k.values())))" >> $OUT
  tail -1 $OUT
done
