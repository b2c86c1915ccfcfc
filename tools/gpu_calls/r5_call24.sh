#!/bin/bash
# round 5, call 24: one COMPLETE PC-sampler run through sampling.get_sampling_fn (N = 1000, 2000 NFE, batch 256) on the final sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 400 python tools/full_sampler_run.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/r5w_full_sampler_run.txt
SSDE_MATRIX=f32 timeout 400 python tools/full_sampler_run.py 256 2>&1 | grep -v amdgpu.ids | sed 's/^/SSDE_MATRIX=f32: /' | tee -a $OUT/r5w_full_sampler_run.txt
