#!/bin/bash
# round 4, call 10: conv_wino4's matrix phase on the BF16 pipe, in isolation (would the split pay there?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for i in 1 2; do timeout 120 tools/microbench/wino4_bf16x6_phase; done > $OUT/r4j_w4x6_phase.txt 2>&1; cat $OUT/r4j_w4x6_phase.txt
