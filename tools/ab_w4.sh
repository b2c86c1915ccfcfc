#!/bin/bash
# Runs ON THE GPU BOX: conv_wino4 trace variants under tools/variants/ (cycle traces) and their timings
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/w4_ab.txt
: > $OUT
for V in $ROOT/tools/variants/*.so; do
  echo "== $(basename $V)" >> $OUT
  SSDE_LIB_PATH=$V timeout 120 python $ROOT/tools/wino4_trace.py 2>&1 | grep -v amdgpu | grep -v "st[1-7]:" >> $OUT
done
cat $OUT
