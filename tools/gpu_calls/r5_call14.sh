#!/bin/bash
# round 5, call 14 (second session): does the Infinity Cache keep a written-then-read window on-die (tools/microbench/mall_window.hip),
# the two-kernel F(4x4,3x3) layers cut into batch chunks on one V window (tools/w4r_chunks.py), and a per-op dump of the current sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 120 tools/microbench/mall_window > $OUT/r5m_mall_window.txt 2>&1; tail -45 $OUT/r5m_mall_window.txt
timeout 200 python tools/w4r_chunks.py 256 > $OUT/r5m_w4r_chunks.txt 2> $OUT/r5m_w4r_chunks.err; cat $OUT/r5m_w4r_chunks.txt; tail -5 $OUT/r5m_w4r_chunks.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-matrix --train-steps 30 --train-warmup 5 --dump-ops $OUT/r5m_ops.json > $OUT/r5m_bench.json 2> $OUT/r5m_bench.err
tail -c 2500 $OUT/r5m_bench.json
