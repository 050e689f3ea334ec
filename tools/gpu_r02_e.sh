#!/bin/bash
# Runs ON THE GPU BOX: GPU suite after the flat-parameter storage; small-batch benches; kernel census of a graph replay.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c4_25k_eager.json 2> $OUT/bench_c4_25k_eager.err
python bench.py --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --graph > $OUT/bench_c4_25k_graph.json 2> $OUT/bench_c4_25k_graph.err
python bench.py --workload c3 --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c3_25k_eager.json 2> $OUT/bench_c3_25k_eager.err
python bench.py --workload c3 --per-gpu 25000 --steps 200 --warmup 20 --no-cpu-baseline --graph > $OUT/bench_c3_25k_graph.json 2> $OUT/bench_c3_25k_graph.err
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/graph_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph_trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --per-gpu 25000 --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline --graph > $OUT/graph_trace/bench.json 2> $OUT/graph_trace/err.txt || echo "rocprof graph trace failed"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,os,csv
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02e"
for f in sorted(glob.glob(root+"/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.3e ms/step %.3f" % (j["value"], j["ms_per_step"]), {k: round(v,3) for k,v in j.get("kernel_ms_per_step",{}).items()})
    except Exception as e:
        print(f, "ERR", e)
for f in glob.glob(root+"/graph_trace/*/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    tot=sum(int(r["Calls"]) for r in rows)
    print("kernel launches in the traced run:", tot, "distinct", len(rows))
    for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:25]:
        print("  %6s calls %9.1f us avg  %s" % (r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
