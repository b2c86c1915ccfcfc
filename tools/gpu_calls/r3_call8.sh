#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" > $OUT/r3h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3h_pytest.log
for M in 4 2; do echo "== SSDE_WGRAD_WINOGRAD=$M"; SSDE_WGRAD_WINOGRAD=$M timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"; done | tee $OUT/r3h_wgrad_bench.txt
