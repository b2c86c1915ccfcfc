#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python tools/interleave_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/r4l_interleave.txt; cat $OUT/r4l_interleave.txt
