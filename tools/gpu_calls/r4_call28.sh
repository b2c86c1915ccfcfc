#!/bin/bash
# round 4, call 28: conv_wino4g_kernel stage body without DMA pieces / without fragment reads (timing experiments)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
{
for V in product w4gnodma w4gnoread w4gneither; do
  if [ $V = product ]; then unset SSDE_LIB_PATH; else export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$V.so; fi
  W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n' | sed -e 's/.*two kernels/  two kernels/' | sed -e "1s/^/$V\n/"
done
} > $OUT/r4aa_w4g_experiments.txt; cat $OUT/r4aa_w4g_experiments.txt
