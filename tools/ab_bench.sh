#!/bin/bash
# Runs ON THE GPU BOX: times tools/conv_bench.py with the product library and with every variant library under
# tools/variants/ (built by score_sde_pytorch_amd/_build.build_variant), same process layout, back to back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}
OUT=$ROOT/gpurun_out
echo "== product library" > $OUT/conv_ab_$TAG.txt
python $ROOT/tools/conv_bench.py 256 >> $OUT/conv_ab_$TAG.txt 2>&1
for V in $ROOT/tools/variants/*.so; do
  case $V in *trace*) continue;; esac
  echo "== variant $(basename $V)" >> $OUT/conv_ab_$TAG.txt
  SSDE_LIB_PATH=$V python $ROOT/tools/conv_bench.py 256 >> $OUT/conv_ab_$TAG.txt 2>&1
done
cat $OUT/conv_ab_$TAG.txt
