#!/bin/bash
# round 5, call 25: the complete PC-sampler run in the headline's matrix mode (SSDE_MATRIX=bf16x6; call 24 ran the library default, f32, twice)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SSDE_MATRIX=bf16x6 timeout 400 python tools/full_sampler_run.py 256 2>&1 | grep -v amdgpu.ids | sed 's/^/SSDE_MATRIX=bf16x6: /' | tee $OUT/r5x_full_sampler_run_bf16x6.txt
