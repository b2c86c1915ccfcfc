#!/bin/bash
# round 6, call 14: per-launch times of one U-Net evaluation (what to look at next)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python tools/op_times.py 256 2>&1 | grep -v amdgpu.ids > $OUT/r6n_op_times.txt
tail -60 $OUT/r6n_op_times.txt
