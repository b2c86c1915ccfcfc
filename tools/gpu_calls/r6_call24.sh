#!/bin/bash
# round 6, call 24: 64-row tiles for the bf16x6 GEMM launches that would not give every CU a 128-row workgroup: parity, shapes, bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_bench_sizes_gpu.py -x -q 2>&1 | flt | tail -4 | tee $OUT/r6o_bm64_rule_parity.txt
timeout 600 python tools/gemm_wide_ab.py 2>&1 | flt | tee $OUT/r6o_gemm_shapes.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6o_bench_err.txt > $OUT/r6o_bench_short.json
python -c "
import json
d = json.load(open('$OUT/r6o_bench_short.json'))
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  1x1 %.2f ms  3x3 %.2f ms  gn %.2f attention %.2f' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv1x1_gemm']['ms'], c['conv3x3_fused']['ms'], c['groupnorm_stats']['ms'], c['attention']['ms']))
" | tee $OUT/r6o_bench_short.txt
