#!/bin/bash
# round 5, call 9: where the 33 k cycles of the F(4x4,3x3) epilogue go (s_memtime stamps per phase)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r5h_wino4r_epilogue_trace.txt
