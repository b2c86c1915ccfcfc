#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py tests/test_sampler_gpu.py -m gpu -x -q -k "winograd or unet or trajectory or one_rank or discrete" > $OUT/r3c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3c_pytest.log
timeout 600 bash tools/ab_bench.sh r3c > /dev/null 2>&1; grep -v amdgpu.ids $OUT/conv_ab_r3c.txt | grep "==\|gn=1"
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4trace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu > $OUT/r3c_w4_trace_full.txt
head -22 $OUT/r3c_w4_trace_full.txt
