#!/bin/bash
# round 6, call 19: the 128 x 256 tile of the bf16x6 GEMM (SSDE_X6_WIDE): parity, per-shape A/B against the 128 x 128 tile and the
# persistent pipelined kernel, sampler / training step A-B-A-B (default rule against SSDE_X6_WIDE=0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -x -q -k "wide or conv1x1_gemm or load_ahead" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/r6j_wide_parity.txt
timeout 600 python tools/gemm_wide_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6j_gemm_wide_ab.txt
F=$OUT/r6j_wide_bench_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
t = d['train'].get('by_class', {})
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  1x1 %.2f ms (train fwd+dgrad %.2f)  3x3 %.2f ms  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv1x1_gemm']['ms'], t.get('conv1x1_fwd+dgrad', {}).get('ms', -1), c['conv3x3_fused']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for V in 0 default; do
    [ $V = default ] && unset SSDE_X6_WIDE || export SSDE_X6_WIDE=$V
    echo "== SSDE_X6_WIDE=$V, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6j_bench_err_$V.txt | line >> $F
  done
done
unset SSDE_X6_WIDE
cat $F
