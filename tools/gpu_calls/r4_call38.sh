#!/bin/bash
# round 4, call 38: the extended dropout-mask test and the two-kernel tests on the GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 200 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "dropout or two_kernels or backward_kernels" 2>&1 | tail -3
