#!/bin/bash
# round 6, call 8: rocprofv3 kernel-trace stats + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the round's kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_gpu.sh r6 2>&1 | tail -30
ls -la $ROOT/gpurun_out/prof_r6 | head; ls -la $ROOT/gpurun_out/profile_summary_r6.json
