#!/bin/bash
# round 5, call 10: the same stamps for a workgroup of the second and of the last round of a launch (warm chip)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for BID in 300 700; do W4R_TRACE_BID=$BID SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/r5i_wino4r_later_workgroups_trace.txt
