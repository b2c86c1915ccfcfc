#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 bash tools/ab_bench.sh r3d > /dev/null 2>&1; grep -v amdgpu.ids $OUT/conv_ab_r3d.txt | grep "==\|gn=1"
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4trace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu > $OUT/r3d_w4_trace_full.txt
head -12 $OUT/r3d_w4_trace_full.txt
