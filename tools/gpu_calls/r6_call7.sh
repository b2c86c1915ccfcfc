#!/bin/bash
# round 6, call 7: the whole GPU suite on the round's kernels, smoke, the default bench line (lease 5 of the telemetry table)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r6g_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/r6g_smoke.txt
timeout 900 python bench.py > $OUT/r6g_bench_default.json 2> $OUT/r6g_bench_default.err
tail -c 1500 $OUT/r6g_bench_default.json; tail -12 $OUT/r6g_bench_default.err
