#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_bench_sizes_gpu.py -m gpu -x -q -k "split_reduction or forward_batch256" > $OUT/r3y_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/r3y_pytest.log
