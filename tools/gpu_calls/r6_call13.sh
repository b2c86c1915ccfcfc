#!/bin/bash
# round 6, call 13: GroupNorm partials merged by the consuming transform pass (ABI 10, gn_in_part0): parity on the GPU, then A-B-A-B against one finalize launch per tensor
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "groupnorm" 2>&1 | flt | tail -6 | tee $OUT/r6m_gn_pass_parity.txt
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_bench_sizes_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | flt | tail -6 | tee -a $OUT/r6m_gn_pass_parity.txt
F=$OUT/r6m_gn_pass_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
g = c.get('groupnorm_stats', {})
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms 1x1 %.2f ms gn %.2f ms (%d launches; %s merged by a pass) sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], c['conv1x1_gemm']['ms'], g.get('ms',0), g.get('launches',0), g.get('merged_by_the_consuming_transform_pass'), d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for W in launch consumer; do
    [ $W = consumer ] && unset SSDE_GN_MERGE_IN_CONSUMER || export SSDE_GN_MERGE_IN_CONSUMER=0
    echo "== $W, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6m_err_$W.txt | line >> $F
  done
done
unset SSDE_GN_MERGE_IN_CONSUMER
cat $F
