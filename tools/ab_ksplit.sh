#!/bin/bash
# Runs ON THE GPU BOX: the 4x4 / 8x8 3x3 layers with and without the two-workgroup reduction split (SSDE_CONV_KSPLIT)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ksplit_ab.txt
: > $OUT
for K in 0 1; do
  echo "== SSDE_CONV_KSPLIT=$K" >> $OUT
  SSDE_CONV_KSPLIT=$K CONV_BENCH_SMALL=1 python $ROOT/tools/conv_bench.py 256 2>&1 | grep -v amdgpu >> $OUT
  SSDE_CONV_KSPLIT=$K python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-train --no-roofline 2>/dev/null | tail -1 | cut -c1-150 >> $OUT
done
cat $OUT
