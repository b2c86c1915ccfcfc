#!/bin/bash
# round 4, call 35: s_memtime trace of wgrad4_gemm_kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_wg4trace.so timeout 200 python tools/wgrad4_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/r4ae_wgrad4_trace.txt; cat $OUT/r4ae_wgrad4_trace.txt
