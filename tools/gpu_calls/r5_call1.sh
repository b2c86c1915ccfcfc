#!/bin/bash
# round 5, call 1: the fill-path microbenchmark + a baseline bench of the round-4 sources on this box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 120 tools/microbench/fill_path > $OUT/r5a_fill_path.txt 2>&1; tail -50 $OUT/r5a_fill_path.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r5a_bench_base.json 2> $OUT/r5a_bench_base.err
tail -c 1500 $OUT/r5a_bench_base.json
