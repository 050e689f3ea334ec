#!/bin/bash
# Development tool: build variants/<name>.so from the working tree with extra hipcc flags for flow.hip (kde/api objects reused).
# Usage: tools/build_variant.sh name [-DFLAG ...]   Prints the fused backward's register / scratch use.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p variants /tmp/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c mentflow_amd/csrc/flow.hip -o /tmp/variants/$NAME.o \
    -Rpass-analysis=kernel-resource-usage 2> /tmp/variants/$NAME.log || { grep -E "error" -A3 /tmp/variants/$NAME.log | head -30; exit 1; }
[ -f mentflow_amd/csrc/api.o ] && [ -f mentflow_amd/csrc/kde.o ] || { echo "build api.o / kde.o first (python __graft_entry__.py)"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$NAME.so mentflow_amd/csrc/api.o mentflow_amd/csrc/kde.o /tmp/variants/$NAME.o
echo -n "$NAME [$*]: fused<20,3> "
grep -A12 "Function Name: _ZN2mf26rqs_layer_bwd_fused_kernelILi20ELi3E" /tmp/variants/$NAME.log | grep -E "VGPRs:|AGPRs:|ScratchSize" | sed "s/.*remark: *//;s/ *\[-R.*//" | tr "\n" ";"; echo
echo -n "   fwd<20,3,1024> "
grep -A12 "Function Name: _ZN2mf20rqs_layer_fwd_kernelILi20ELi3ELi1024E" /tmp/variants/$NAME.log | grep -E "VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed "s/.*remark: *//;s/ *\[-R.*//" | tr "\n" ";"; echo
