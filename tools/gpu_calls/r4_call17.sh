#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "bf16_matrix_pipe" 2>&1 | tail -2
W4_BOUNDS_SPLIT=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > $OUT/r4q_w4x.txt; cat $OUT/r4q_w4x.txt
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4xtrace.so timeout 200 python tools/wino4x_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/r4q_w4x_trace.txt; cat $OUT/r4q_w4x_trace.txt
