#!/bin/bash
# round 6, call 33: per-launch list of one FFHQ-256 evaluation at batch 16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
SSDE_MATRIX=bf16x6 timeout 600 python tools/op_times.py 16 ve/ffhq_256_ncsnpp_continuous 2>&1 | grep -v "amdgpu.ids" > $ROOT/gpurun_out/r6w_op_times_ffhq256.txt
grep -A45 'sums per' $ROOT/gpurun_out/r6w_op_times_ffhq256.txt
