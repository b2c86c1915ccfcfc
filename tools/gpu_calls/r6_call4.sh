#!/bin/bash
# round 6, call 4: residual / addend rows of the F(4x4,3x3) epilogue fetched ahead (first form: a struct behind a nullable pointer) -- layer A/B and sampler / train A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
V=$ROOT/tools/variants/libssde_hip_w4rnopf.so
F=$OUT/r6d_w4r_epilogue_prefetch_ab.txt
: > $F
for rep in 1 2; do
  echo "== product (rows fetched ahead), pass $rep" >> $F
  timeout 300 python tools/w4r_resid_ab.py 2>&1 | grep -v amdgpu.ids >> $F
  echo "== variant w4rnopf (store phase fetches them), pass $rep" >> $F
  SSDE_LIB_PATH=$V timeout 300 python tools/w4r_resid_ab.py 2>&1 | grep -v amdgpu.ids >> $F
done
for rep in 1 2; do
  echo "== product, bench pass $rep" >> $F
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 class %.2f ms  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], d['roofline']['by_class']['conv3x3_fused']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))" >> $F
  echo "== variant w4rnopf, bench pass $rep" >> $F
  SSDE_LIB_PATH=$V timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 class %.2f ms  sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], d['roofline']['by_class']['conv3x3_fused']['ms'], d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))" >> $F
done
cat $F
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix > $OUT/r6d_bench_short.json 2> $OUT/r6d_bench_short.err
tail -c 3000 $OUT/r6d_bench_short.json
