#!/bin/bash
# Runs ON THE GPU BOX: KDE kernel tests + tuning sweeps (1-D and 2-D) after the factorised-window rewrite.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_golden.py tests/test_full_size_properties.py tests/test_baseline_configs.py tests/test_flow_kernels.py tests/test_fused_backward.py -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python tools/kde_sweep.py --sweep 1d > $OUT/sweep_1d.txt 2>&1 && echo "1d sweep done" &&
python tools/kde_sweep.py --sweep 2d > $OUT/sweep_2d.txt 2>&1 && echo "2d sweep done"
sort -t'"' -k1 $OUT/sweep_1d.txt | head -60
cat $OUT/sweep_2d.txt | head -70
