#!/bin/bash
# round 6, call 9: conv_wino4r_kernel with persistent workgroups (next tile's first loads issued from the epilogue) against one
# workgroup per tile: parity first, then layers and sampler / train A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_bench_sizes_gpu.py tests/test_unet_gpu.py -x -q -k "winograd or bench or cifar or forward or trajectory or iteration or gradients" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/r6i_persistent_parity.txt
F=$OUT/r6i_w4r_persistent_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms (frac %.3f)  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], d['roofline']['frac'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for V in product w4rnopersist; do
    [ $V = product ] && unset SSDE_LIB_PATH || export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$V.so
    echo "== $V, layers pass $rep" >> $F
    timeout 300 python tools/w4r_resid_ab.py 2>&1 | grep -v amdgpu.ids >> $F
    echo "== $V, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>/dev/null | line >> $F
  done
done
unset SSDE_LIB_PATH
cat $F
