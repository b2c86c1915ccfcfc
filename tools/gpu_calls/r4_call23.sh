#!/bin/bash
# round 4, call 23: conv_wino4_kernel without prologue + input transform in the stage body (upper bound of a two-kernel Winograd main loop)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
{
for i in 1 2; do
  timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids
  SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4noxform.so timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids
done
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4noxformtrace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu.ids | head -16
} > $OUT/r4v_w4_noxform.txt
cat $OUT/r4v_w4_noxform.txt
