#!/bin/bash
# round 4, call 11: F(4x4,3x3) on 4x4 maps (32 whole images per workgroup, interior-only staging, reduction split over 4) vs direct
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for b in 256 128; do CONV_BENCH_SHAPES="256,256,4;384,256,4;256,128,4" timeout 300 python tools/conv_bench.py $b; done 2>&1 | grep -v amdgpu.ids > $OUT/r4k_conv4x4.txt; cat $OUT/r4k_conv4x4.txt
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "winograd_f4x4" 2>&1 | tail -2
