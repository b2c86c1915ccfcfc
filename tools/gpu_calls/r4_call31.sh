#!/bin/bash
# round 4, call 31: profiles of the final sources (kernel trace + PMC passes, sampler and training step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_gpu.sh r4final 2>&1 | tail -15
bash tools/profile_train_pmc.sh r4final 2>&1 | tail -5
