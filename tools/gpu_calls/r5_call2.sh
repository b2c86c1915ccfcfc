#!/bin/bash
# round 5, call 2: the fill-path microbenchmark; call 1 had no binary (baseline bench only: 4.758 img/s, 0.05806 s/step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 150 tools/microbench/fill_path > $OUT/r5a_fill_path.txt 2>&1; tail -60 $OUT/r5a_fill_path.txt
