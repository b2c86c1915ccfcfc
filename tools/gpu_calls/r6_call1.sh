#!/bin/bash
# round 6, call 1: the reference-pinned training / checkpoint tests on the MI355X + the default bench line (baseline of the round)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -k "reference" 2>&1 | tail -15 | tee $OUT/r6a_train_reference_tests.txt
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/r6a_bench_default.txt
