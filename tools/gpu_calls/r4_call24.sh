#!/bin/bash
# round 4, call 24: F(4x4,3x3) in two kernels (conv_wino4g.hip): parity, then the transform pass and the matrix kernel timed apart
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels" 2>&1 | tail -3
for i in 1 2; do W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n'; done > $OUT/r4w_two_kernels.txt; cat $OUT/r4w_two_kernels.txt
