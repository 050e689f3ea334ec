#!/bin/bash
# Development tool: build variants/<name>.so from the working tree with extra hipcc flags for ONE translation unit of
# mentflow_amd/csrc/SOURCES.txt (default: flow_bwd_fused_s0; VARIANT_TU=flow_fwd ... selects another; the other objects are
# reused from the last __graft_entry__.build()).  Usage: [VARIANT_TU=obj] tools/build_variant.sh name [-DFLAG ...]
# Prints the register / scratch use of the main instances.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
TU=${VARIANT_TU:-flow_bwd_fused_s0}
CS=mentflow_amd/csrc
mkdir -p variants /tmp/variants
LINE=$(grep "^$TU " $CS/SOURCES.txt) || { echo "no object $TU in SOURCES.txt"; exit 1; }
set -- $LINE "$@"; shift; SRC=$1; shift
OBJS=""
for o in $(grep -v '^#' $CS/SOURCES.txt | awk 'NF {print $1}'); do
  if [ "$o" = "$TU" ]; then OBJS="$OBJS /tmp/variants/$NAME.o"; else
    [ -f $CS/$o.o ] || { echo "build $CS/$o.o first (python __graft_entry__.py)"; exit 1; }
    OBJS="$OBJS $CS/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $CS/$SRC -o /tmp/variants/$NAME.o \
    -Rpass-analysis=kernel-resource-usage 2> /tmp/variants/$NAME.log || { grep -E "error" -A3 /tmp/variants/$NAME.log | head -30; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$NAME.so $OBJS
echo "$NAME [$TU: $*]"
python3 - /tmp/variants/$NAME.log <<'PY'
import re, sys
cur, row = None, {}
for line in open(sys.argv[1]):
    m = re.search(r'remark: (.*)', line)
    if not m: continue
    t = m.group(1).split("[-Rpass")[0].strip()
    if t.startswith('Function Name:'): cur = t.split(':', 1)[1].strip(); row = {}
    for key in ('VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs'):
        if t.startswith(key + ':'): row[key] = t.split(':')[1].strip()
    if t.startswith('LDS Size') and ('ILi20ELi3E' in cur):
        print(f"   {cur[:60]:60s} v{row.get('VGPRs')} a{row.get('AGPRs')} s{row.get('SGPRs')} scratch {row.get('ScratchSize [bytes/lane]')} occ {row.get('Occupancy [waves/SIMD]')}")
PY
