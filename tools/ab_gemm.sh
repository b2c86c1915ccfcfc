#!/bin/bash
# Runs ON THE GPU BOX: tools/gemm_bench.py with the product library and every variant under tools/variants/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}
OUT=$ROOT/gpurun_out/gemm_ab_$TAG.txt
echo "== product library" > $OUT
python $ROOT/tools/gemm_bench.py 256 >> $OUT 2>&1
for V in $ROOT/tools/variants/*.so; do
  case $V in *trace*) continue;; esac
  echo "== variant $(basename $V)" >> $OUT
  SSDE_LIB_PATH=$V python $ROOT/tools/gemm_bench.py 256 >> $OUT 2>&1
done
grep -v amdgpu $OUT
