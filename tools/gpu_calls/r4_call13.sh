#!/bin/bash
# round 4, call 13: rocprofv3 kernel-trace stats + PMC passes of the bench command on the final sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_gpu.sh r4final 2>&1 | tail -25
bash tools/profile_train_pmc.sh r4 2>&1 | tail -5
