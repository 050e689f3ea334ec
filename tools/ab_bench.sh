#!/bin/bash
# Development tool: A/B a compile-time flag.  Rebuilds ONE translation unit (AB_TU, default flow_bwd_fused_s0) of the library
# with extra hipcc flags (arg 1, may be empty), runs bench.py with the remaining args as env assignments, prints the kernel times.
# Usage: tools/ab_bench.sh "-DSOME_FLAG" MENTFLOW_BWD_FUSED=0
# NOTE: overwrites mentflow_amd/csrc/libmentflow_hip.so in the working copy it runs in (a scratch copy under gpurun);
# rebuild with __graft_entry__.build() afterwards when used locally.
set -e
FLAGS="$1"; shift
cd "$(dirname "$0")/.."
VARIANT_TU=${AB_TU:-flow_bwd_fused_s0} tools/build_variant.sh _ab $FLAGS > /dev/null
cp variants/_ab.so mentflow_amd/csrc/libmentflow_hip.so
env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$FLAGS $*', round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['kernel_ms_per_step'].items()})"
