#!/bin/bash
# round 6, call 3: segmented-graph exchange step + lazy device weight packing (GPU suite subset), FIR staging A/B,
# RCCL two ranks on one device, a short bench line with the one-rank exchange probe
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | tail -6 | tee $OUT/r6c_gpu_tests.txt
echo "== product (kStageU = 4)" > $OUT/r6c_fir_ab.txt
timeout 300 python tools/fir_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT/r6c_fir_ab.txt
echo "== variant firu1 (one load in flight per thread: rounds 2-5)" >> $OUT/r6c_fir_ab.txt
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_firu1.so timeout 300 python tools/fir_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT/r6c_fir_ab.txt
cat $OUT/r6c_fir_ab.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_two_ranks_one_device.py 2>&1 | grep -v amdgpu.ids | tail -20 | tee $OUT/r6c_rccl_two_ranks_one_device.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/r6c_bench_short.txt
