#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "bf16_matrix_pipe" 2>&1 | tail -2
W4_BOUNDS_SPLIT=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > $OUT/r4o_w4x.txt; cat $OUT/r4o_w4x.txt
