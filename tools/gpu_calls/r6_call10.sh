#!/bin/bash
# round 6, call 10: the persistent conv_wino4r form with hazard-safe prefetch loads (variant library): parity, then A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
V=$ROOT/tools/variants/libssde_hip_w4rpersist.so
SSDE_LIB_PATH=$V timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "winograd_f4x4" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/r6j_persistent_parity.txt
SSDE_LIB_PATH=$V timeout 1200 python -m pytest tests/test_bench_sizes_gpu.py tests/test_unet_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee -a $OUT/r6j_persistent_parity.txt
F=$OUT/r6j_w4r_persistent_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms (frac %.3f)  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], d['roofline']['frac'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for W in product w4rpersist; do
    [ $W = product ] && unset SSDE_LIB_PATH || export SSDE_LIB_PATH=$V
    echo "== $W, layers pass $rep" >> $F
    timeout 300 python tools/w4r_resid_ab.py 2>&1 | grep -v amdgpu.ids | tail -6 >> $F
    echo "== $W, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>/dev/null | line >> $F
  done
done
unset SSDE_LIB_PATH
cat $F
