#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round-2 baseline — GPU tests, default bench, 2-rank rehearsal, strong-scaling N=1
# point (16 M particles), C5 bench + C5 rocprofv3 stats / SQ counters, fused-backward cycle stamps.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
./tools/bin/ubench_lds_atomics2 > $OUT/ubench_lds_atomics2.txt 2>&1 || echo "ubench failed"
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -30 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench default done" &&
MENTFLOW_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err && echo "2 ranks done" &&
python bench.py --scaling strong --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/bench_strong_n1.json 2> $OUT/bench_strong_n1.err && echo "strong done" &&
python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err && echo "c5 done" &&
python tools/fb_diag.py > $OUT/fb_diag.txt 2>&1 && echo "diag done"
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/c5_stats $OUT/c5_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > $OUT/c5_stats/bench.json 2> $OUT/c5_stats/err.txt && echo "c5 stats done" &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/c5_sq -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2> $OUT/c5_sq/err.txt && echo "c5 sq done"
cd $GRAFT_REPO_ROOT
for f in bench_default bench_2ranks_shared bench_strong_n1 bench_c5; do echo "== $f"; tail -c 1800 $OUT/$f.json; echo; done
cat $OUT/fb_diag.txt | tail -20
