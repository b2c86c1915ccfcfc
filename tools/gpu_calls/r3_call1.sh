#!/bin/bash
# Runs ON THE GPU BOX: parity first, then the F(4x4,3x3) A/B (product vs the round-2 kernel vs variants), traces, counters, bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3a_pytest.log
timeout 600 bash tools/ab_bench.sh r3a > /dev/null 2>&1; grep -v amdgpu.ids $OUT/conv_ab_r3a.txt | grep "==\|gn=1" 
timeout 300 bash tools/ab_w4.sh > /dev/null 2>&1; cp $OUT/w4_ab.txt $OUT/r3a_w4_trace.txt; head -30 $OUT/w4_ab.txt
SHAPE="128 128 32 1" timeout 600 bash tools/w4_pmc.sh > $OUT/r3a_w4_pmc.txt 2>&1; tail -40 $OUT/r3a_w4_pmc.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/r3a_bench.json 2> $OUT/r3a_bench.err; echo "bench rc=$?"; cat $OUT/r3a_bench.json | cut -c1-600
