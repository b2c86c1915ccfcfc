#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
V=$ROOT/tools/variants/libssde_hip_gn3op.so
(for R in 1 2; do
echo "== product (fma form)"; timeout 200 python tools/conv_bench.py 256 2>&1 | grep "gn=1"
echo "== variant gn3op (sub, mul, fma)"; SSDE_LIB_PATH=$V timeout 200 python tools/conv_bench.py 256 2>&1 | grep "gn=1"
done) | tee $OUT/r3x_gn_fma_ab.txt
for L in product variant product variant; do
  if [ $L = variant ]; then export SSDE_LIB_PATH=$V; else unset SSDE_LIB_PATH; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline --no-train > $OUT/r3x_bench_$L.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/r3x_bench_$L.json")); print("$L sampler ms", d["ms_per_step"], d["value"])
PY
done
