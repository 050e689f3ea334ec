#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats and the HBM-traffic PMC passes of the default
# bench command, written under gpurun_out/profiles_<tag>/ ; tools/summarise_pmc.py turns them into profiles/*.
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
# Usage: collect_profiles.sh <tag> <commit> [all|pmc|benches|stamps]   (each part fits one 20-minute gpurun call; `all` may not)
set -e
TAG=${1:-r01}
PART=${3:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq
part() { [ "$PART" = "all" ] || [ "$PART" = "$1" ]; }
finish() {
  # gpurun merges at most 64 MiB back: keep the summaries the tools read, drop the raw traces and the diagnostic library
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*domain_stats.csv" -delete
  rm -f $GRAFT_REPO_ROOT/gpurun_out/libmentflow_diag.so $GRAFT_REPO_ROOT/gpurun_out/diag_*.o
  echo "$COMMIT" > $OUT/commit.txt
  # condense ON THE BOX (tools/summarise_pmc.py writes profiles/<tag>_* in this scratch copy of the repo) and hand back only the
  # summaries: the raw rocprofv3 counter tables run to tens of MB per pass
  cd $GRAFT_REPO_ROOT && python3 tools/summarise_pmc.py $TAG > /dev/null
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles_out
  cp $GRAFT_REPO_ROOT/profiles/${TAG}_* $GRAFT_REPO_ROOT/gpurun_out/profiles_out/
  [ -f $OUT/fetch_done ] && cp $GRAFT_REPO_ROOT/profiles/traffic.json $GRAFT_REPO_ROOT/gpurun_out/profiles_out/
  rm -rf $OUT
}
COMMIT=${2:-unknown}
if part pmc; then
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-strong-n1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats/bench.json 2> $OUT/stats/err.txt
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > /dev/null 2> $OUT/fetch/err.txt
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > /dev/null 2> $OUT/write/err.txt
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > /dev/null 2> $OUT/sq/err.txt
echo "sq done"; touch $OUT/fetch_done
# C5 (2-D KDE dominant after the flow): kernel stats + SQ counters
mkdir -p $OUT/c5_stats $OUT/c5_sq
BENCH5="python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- $BENCH5 > $OUT/c5_stats/bench.json 2> $OUT/c5_stats/err.txt
echo "c5 stats done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/c5_sq -- $BENCH5 > /dev/null 2> $OUT/c5_sq/err.txt
echo "c5 sq done"
fi
if part benches; then
cd $GRAFT_REPO_ROOT && python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --scaling strong --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_strong_n1.json 2> $OUT/bench_strong_n1.err
python3 bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
MENTFLOW_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline 2> $OUT/bench_2ranks_shared.err | grep "^{" > $OUT/bench_2ranks_shared.json
for w in c1 c2 c3; do
  python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
python3 bench.py --workload c3 --per-gpu 25000 --steps 50 --warmup 5 --no-cpu-baseline --meas-samples 200000 > $OUT/bench_c3_25k_eager.json 2> $OUT/bench_c3_25k_eager.err
python3 bench.py --workload c3 --per-gpu 25000 --steps 50 --warmup 5 --no-cpu-baseline --meas-samples 200000 --graph --fused-adamw > $OUT/bench_c3_25k_graph_fused.json 2> $OUT/bench_c3_25k_graph_fused.err
echo "other workloads done"
fi
if part stamps; then
cd $GRAFT_REPO_ROOT
# in-kernel cycle stamps of the fused backward (diagnostic build) and the timing ablations
for lv in 0 1 2; do
  FB_DIAG_LEVEL=$lv python3 tools/fb_diag.py 2>/dev/null >> $OUT/fused_bwd_cycles.txt
done
for abl in NO_BARRIER NO_DW NO_SPLINE; do
  FB_DIAG_LEVEL=2 WS_DIAG_FLAGS="-DMF_FB_$abl" python3 tools/fb_diag.py 2>/dev/null | head -18 > $OUT/fused_bwd_ablation_$(echo $abl | tr A-Z a-z).txt
done
# the hand-off levels side by side (one bench line each) and the batch sweep of the strong-scaling series
for lv in 0 1 2; do
  MENTFLOW_ACT_LEVEL=$lv python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-strong-n1 > $OUT/bench_level$lv.json 2> $OUT/bench_level$lv.err
done
echo "stamps done"
fi
finish
