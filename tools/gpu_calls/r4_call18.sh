#!/bin/bash
# is the epilogue's store burst HBM-bound because every workgroup stores at the same time?  the same trace with 4 workgroups on the chip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for B in 256 1; do
  echo "### batch $B"
  W4_TRACE_BATCH=$B SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4trace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu.ids | grep -v "   st"
done > $OUT/r4r_w4_trace_batch.txt
cat $OUT/r4r_w4_trace_batch.txt
