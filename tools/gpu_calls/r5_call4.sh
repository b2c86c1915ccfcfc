#!/bin/bash
# round 5, call 4: conv_wino4r with two 4-wave workgroups per CU (the second one started half a tile late) against the 8-wave form
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 60 tools/microbench/wg_slots > $OUT/r5c_wg_slots.txt 2>&1; head -100 $OUT/r5c_wg_slots.txt | cut -c1-200
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels or register_fed or repack or dropout" 2>&1 | tail -4
{
W4_BOUNDS_TWO=1 timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids
for ST in 1 2 3 0; do echo "== SSDE_DEBUG=w4r_stagger=$ST"; SSDE_DEBUG=w4r_stagger=$ST W4_BOUNDS_TWO=r timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids; done
for ST in 0 1; do echo "== trace, stagger $ST"; SSDE_DEBUG=w4r_stagger=$ST SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids; done
} > $OUT/r5c_wino4r_narrow.txt 2>&1
cat $OUT/r5c_wino4r_narrow.txt
