#!/bin/bash
# round 6, call 20: life of every workgroup of a bf16x6 GEMM launch (128 x 128 against 128 x 256 tile), and the stage trace of
# workgroup 0; then the whole GPU suite on the new rule
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_gemmtrace.so
timeout 300 python tools/gemm_wg_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6k_gemm_wg_trace.txt
unset SSDE_LIB_PATH
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/r6k_gpu_suite.txt
