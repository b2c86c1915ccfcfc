#!/bin/bash
# per-stage phases of conv_wino4_kernel with the chip full (batch 256: 512 workgroups) and nearly empty (batch 1, 2, 8, 32, 64)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for B in 256 64 32 8 1; do
  echo "### batch $B"
  W4_TRACE_BATCH=$B SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4trace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu.ids | head -24 | grep -v "wave 7" | head -13
done > $OUT/r4r_w4_trace_batch2.txt
cat $OUT/r4r_w4_trace_batch2.txt
