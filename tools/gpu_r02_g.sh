#!/bin/bash
# Runs ON THE GPU BOX: timing ablations of the fused backward (diagnostic builds; results of the ablated builds are wrong).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in base NO_BARRIER NO_DW NO_SPLINE; do
  if [ $v = base ]; then export WS_DIAG_FLAGS=""; else export WS_DIAG_FLAGS="-DMF_FB_$v"; fi
  python tools/fb_diag.py > $OUT/fb_diag_$v.txt 2>&1 || echo "diag $v failed"
  echo "=== $v"; grep -E "TOTAL|barrier A|trunk fwd|phi|rqs_apply|dW last|gh \+=|trunk bwd|w0:" $OUT/fb_diag_$v.txt
done
