#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 bash tools/profile_train_pmc.sh r3 2>&1 | tail -30
