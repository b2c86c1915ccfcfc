#!/bin/bash
# round 6, call 17: attention forward alone, both kernels (four-stage register ring with counted waits) + its tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6q_attn_bench.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -5 | tee -a $OUT/r6q_attn_bench.txt
