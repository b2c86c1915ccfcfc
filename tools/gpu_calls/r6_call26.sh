#!/bin/bash
# round 6, call 26: per-launch list of the training program (forward + backward) at batch 128
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SSDE_MATRIX=bf16x6 timeout 900 python tools/train_op_times.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/r6q_train_op_times.txt
tail -60 $OUT/r6q_train_op_times.txt
