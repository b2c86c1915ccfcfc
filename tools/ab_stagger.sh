#!/bin/bash
# Runs ON THE GPU BOX: 1x1 GEMM micro-benchmark and the 3x3 micro-benchmark with a residual input, product library
# against every variant under tools/variants/, then the sampler bench line for each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stagger_ab.txt
: > $OUT
for V in product $ROOT/tools/variants/*.so; do
  [ $V = product ] && unset SSDE_LIB_PATH || export SSDE_LIB_PATH=$V
  echo "== $(basename $V): gemm" >> $OUT
  python $ROOT/tools/gemm_bench.py 256 2>&1 | grep -v amdgpu >> $OUT
  case $V in product|*rpref.so)
    echo "== $(basename $V): 3x3 + residual" >> $OUT
    CONV_BENCH_RESID=1 python $ROOT/tools/conv_bench.py 256 2>&1 | grep -v amdgpu >> $OUT;;
  esac
  echo "== $(basename $V): sampler" >> $OUT
  python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-train --no-roofline 2>/dev/null | tail -1 | cut -c1-140 >> $OUT
done
cat $OUT
