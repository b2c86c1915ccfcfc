#!/bin/bash
# round 4, call 29: the whole GPU suite on the current sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r4ab_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r4ab_pytest_gpu.log
