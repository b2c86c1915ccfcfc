#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4xtrace.so timeout 200 python tools/wino4x_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/r4p_w4x_trace.txt; cat $OUT/r4p_w4x_trace.txt
