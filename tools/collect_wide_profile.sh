#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats and the HBM-traffic / SQ counter passes of the wide conditioner family
# (tools/bench_wide.py: the C4 problem with 128 hidden units, 1 048 576 particles) into gpurun_out/wide_<tag>/; afterwards, here:
#   python tools/summarise_wide.py <tag>    ->  profiles/<tag>_wide128_{kernels.csv,bench.json}
# Counters in their own passes, never together with a trace domain other than --kernel-trace (gpurun's rule).
set -e
TAG=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out/wide_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/bench_wide.py --hidden-units 128 --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench.json 2> $O/err_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/err_fetch.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/err_write.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $B > /dev/null 2> $O/err_sq.txt
echo "collected $O"
