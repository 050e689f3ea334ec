#!/bin/bash
# Development tool: per-kernel register / spill / LDS / occupancy report of one translation unit (default flow_bwd_fused.hip; or $1, extra flags after it) from hipcc's own remarks.
SRC=${1:-mentflow_amd/csrc/flow_bwd_fused.hip}; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$SRC" -o /dev/null --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r'remark: (.*)', line)
    if not m: continue
    t = m.group(1).split("[-Rpass")[0].strip()
    if t.startswith('Function Name:'): cur = t.split(':',1)[1].strip(); row = {}
    for key in ('VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs', 'VGPRs Spill', 'SGPRs Spill', 'LDS Size [bytes/block]'):
        if t.startswith(key + ':'): row[key] = t.split(':')[1].strip()
    if t.startswith('LDS Size'):
        print(f\"{cur[:70]:70s} v{row.get('VGPRs')} a{row.get('AGPRs')} s{row.get('SGPRs')} scratch {row.get('ScratchSize [bytes/lane]')} occ {row.get('Occupancy [waves/SIMD]')}\")
"
