#!/bin/bash
# round 6, call 22: the GroupNorm merge without contraction (one function, the same bits in every kernel it is inlined into):
# the test that failed on a clean build, the whole suite, a short bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 600 python -m pytest tests/test_train_gpu.py -q -k "groupnorm" 2>&1 | flt | tail -4 | tee $OUT/r6m_gn_merge_no_contract.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | flt | tail -8 | tee $OUT/r6m_gpu_suite.txt
timeout 900 python bench.py --no-cpu-baseline --no-extras --no-exchange-probe > $OUT/r6m_bench_short.json 2>$OUT/r6m_bench_err.txt
python -c "
import json
d = json.load(open('$OUT/r6m_bench_short.json'))
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  1x1 %.2f ms  3x3 %.2f ms  gn %.2f attention %.2f' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv1x1_gemm']['ms'], c['conv3x3_fused']['ms'], c['groupnorm_stats']['ms'], c['attention']['ms']))
print('f32 mode:', d.get('matrix_f32', {}).get('value'), d.get('matrix_f32', {}).get('train_s_per_step'))
" | tee $OUT/r6m_bench_short.txt
