#!/bin/bash
# round 6, call 6: ssde_store_tile in three straight passes (values, stores, statistics) against the row-by-row loop (a vmcnt(0)
# behind every store), and the residual prefetch of conv_wino4r on top: layers, GEMMs, sampler / train A-B-C-A-B-C
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
F=$OUT/r6f_store_passes_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms  1x1 %.2f ms  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], c['conv1x1_gemm']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for V in product rowbyrow w4rnopf wg4elem; do
    [ $V = product ] && unset SSDE_LIB_PATH || export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$V.so
    echo "== $V, layers pass $rep" >> $F
    timeout 300 python tools/w4r_resid_ab.py 2>&1 | grep -v amdgpu.ids >> $F
    if [ $rep = 1 ]; then
      echo "== $V, 1x1 GEMMs" >> $F
      timeout 300 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu.ids | head -14 >> $F
    fi
    echo "== $V, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>/dev/null | line >> $F
  done
done
unset SSDE_LIB_PATH
cat $F
