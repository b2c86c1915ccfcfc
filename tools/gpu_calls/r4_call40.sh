#!/bin/bash
# round 4, call 40: the training tests on the GPU with the stream-K weight-gradient split as the default
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 150 python -m pytest tests/test_train_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q -k "whole_network or fed_by or fused_training or stream_k or gradients" 2>&1 | tail -3
