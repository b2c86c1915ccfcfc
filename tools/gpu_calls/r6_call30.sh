#!/bin/bash
# round 6, call 30: the long-K, few-rows GEMM of the training step's backward under every existing route
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python tools/gemm_long_k_ab.py 2>&1 | grep -v "amdgpu.ids" | tee $ROOT/gpurun_out/r6t_gemm_long_k.txt
