#!/bin/bash
# round 6, call 15: the attention forward on the BF16 matrix pipe (attn_x6_kernel): parity, distance from fp64 against the fp32 kernel, A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | flt | tail -12 | tee $OUT/r6o_attn_x6_parity.txt
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_attn6.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "as_close_to_fp64" 2>&1 | flt | grep -E "^E   +Assert|passed|failed" | tee -a $OUT/r6o_attn_x6_parity.txt
timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_bench_sizes_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | flt | tail -6 | tee -a $OUT/r6o_attn_x6_parity.txt
F=$OUT/r6o_attn_x6_ab.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
t = d['train'].get('by_class', {})
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  attention %.3f ms (train %.3f)  3x3 %.2f ms  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['attention']['ms'], t.get('attention', {}).get('ms', 0), c['conv3x3_fused']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for W in fp32 x6 x6_six_terms; do
    unset SSDE_ATTN_X6 SSDE_LIB_PATH
    [ $W = fp32 ] && export SSDE_ATTN_X6=0
    [ $W = x6_six_terms ] && export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_attn6.so
    echo "== $W, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6o_err_$W.txt | line >> $F
  done
done
unset SSDE_ATTN_X6 SSDE_LIB_PATH
cat $F
