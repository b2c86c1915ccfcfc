#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
(for W in 384 576 768 1024; do echo "== F(4x4,3x3) SSDE_WGRAD4_WGS=$W"; SSDE_WGRAD_WINOGRAD=44 SSDE_WGRAD4_WGS=$W timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"; done
echo "== F(2x2,3x3)"; SSDE_WGRAD_WINOGRAD=2 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2") | tee $OUT/r3j_wgrad_bench.txt
