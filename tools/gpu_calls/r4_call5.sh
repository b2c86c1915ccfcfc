#!/bin/bash
# round 4, call 5: energy per launch of the candidate kernels (the sampler runs at the power cap); batch sweep of the sampler
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
rocm-smi --showmaxpower --showpower 2>&1 | grep -v amdgpu.ids | head -20 > $OUT/r4e_powercap.txt; cat $OUT/r4e_powercap.txt
timeout 400 python tools/energy_probe.py 256 2>&1 | grep -v amdgpu.ids > $OUT/r4e_energy.txt; cat $OUT/r4e_energy.txt
timeout 900 python tools/batch_sweep.py 16 64 256 2>&1 | grep -v amdgpu.ids > $OUT/r4e_batch_sweep.txt; cat $OUT/r4e_batch_sweep.txt
