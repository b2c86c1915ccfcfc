#!/bin/bash
# round 6, call 23: the transform pass merges the GroupNorm partials of its source BEFORE it requests its pixels (no register of
# the merge live across the 144 pixel registers): parity, then sampler / training step A-B-A-B against the finalize launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 600 python -m pytest tests/test_train_gpu.py -q -k "groupnorm" 2>&1 | flt | tail -4 | tee $OUT/r6n_gn_merge_first_parity.txt
SSDE_GN_MERGE_IN_CONSUMER=1 timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_bench_sizes_gpu.py tests/test_sampler_gpu.py tests/test_train_gpu.py -x -q 2>&1 | flt | tail -4 | tee -a $OUT/r6n_gn_merge_first_parity.txt
F=$OUT/r6n_gn_merge_first_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms  gn %.2f ms (%d launches)  1x1 %.2f  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], c['groupnorm_stats']['ms'], c['groupnorm_stats']['launches'], c['conv1x1_gemm']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for V in 0 1; do
    export SSDE_GN_MERGE_IN_CONSUMER=$V
    echo "== SSDE_GN_MERGE_IN_CONSUMER=$V, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6n_bench_err_$V.txt | line >> $F
  done
done
unset SSDE_GN_MERGE_IN_CONSUMER
cat $F
