#!/bin/bash
# round 6, call 29: evidence on the final sources -- one COMPLETE sampler run (N = 1000) in both matrix modes, a short bench line
# that reads the committed PMC summary (pmc_stale false), the new deferred-finish test on the GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 600 python -m pytest tests/test_train_gpu.py -q -k "deferred_finishing" 2>&1 | flt | tail -3 | tee $OUT/r6s_deferred_finish_test.txt
: > $OUT/r6s_full_sampler_run.txt
for M in bf16x6 f32; do
  echo "== SSDE_MATRIX=$M" >> $OUT/r6s_full_sampler_run.txt
  SSDE_MATRIX=$M timeout 600 python tools/full_sampler_run.py 2>&1 | flt | tail -4 >> $OUT/r6s_full_sampler_run.txt
done
cat $OUT/r6s_full_sampler_run.txt
timeout 900 python bench.py --no-cpu-baseline --no-extras --no-other-matrix > $OUT/r6s_bench_short_final.json 2>$OUT/r6s_bench_short_final.err
python -c "
import json
d = json.load(open('$OUT/r6s_bench_short_final.json'))
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  pmc_stale %s  frac %.3f traffic %.0f MB mfma_busy %.3f' % (d['value'], d['ms_per_step'], d['train']['value'], d['config'].get('pmc_stale'), d['roofline']['frac'], d['roofline']['traffic'] / 1e6, d['roofline']['mfma_busy_pmc']))"
