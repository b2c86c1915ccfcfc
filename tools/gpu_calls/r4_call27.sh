#!/bin/bash
# round 4, call 27: conv_wino4g_kernel with wave-private V (no barrier in the main loop): parity, timing, trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -x -q -k "two_kernels" 2>&1 | tail -2
{
for i in 1 2; do
  W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n'
done
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4gtrace.so timeout 200 python tools/wino4g_trace.py 2>&1 | grep -v amdgpu.ids
} > $OUT/r4z_two_kernels_private_v.txt; cat $OUT/r4z_two_kernels_private_v.txt
