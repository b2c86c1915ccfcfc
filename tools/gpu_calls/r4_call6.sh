#!/bin/bash
# round 4, call 6: energy per launch again, with rocm-smi as the power source
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python tools/energy_probe.py 256 2>&1 | grep -v amdgpu.ids > $OUT/r4f_energy.txt; cat $OUT/r4f_energy.txt
