#!/bin/bash
# round 4, call 25: conv_wino4g_kernel with the stage barrier in the middle of the stage: parity, A/B against the barrier at the end, trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels" 2>&1 | tail -2
{
for i in 1 2; do
  W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n'
  SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4gendbar.so W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n'
done
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4gtrace.so timeout 200 python tools/wino4g_trace.py 2>&1 | grep -v amdgpu.ids
} > $OUT/r4x_two_kernels_midbar.txt; cat $OUT/r4x_two_kernels_midbar.txt
