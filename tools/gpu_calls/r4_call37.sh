#!/bin/bash
# round 4, call 37: non-temporal stores in the shared epilogue (timing experiment): conv kernels and the 1x1 GEMM
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
{
for V in product storent product storent; do
  if [ $V = product ]; then unset SSDE_LIB_PATH; else export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$V.so; fi
  W4_BOUNDS_TWO=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n' | sed -e "1s/^/$V /"
done
for V in product storent product storent; do
  if [ $V = product ]; then unset SSDE_LIB_PATH; else export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$V.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-train > $OUT/r4ag_bench_$V.json 2> $OUT/r4ag_bench_$V.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4ag_bench_$V.json") if x.startswith("{")]
d = json.loads(l[-1])
print("$V", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"].get("by_class", {}).items()})
PY
done
} 2>&1 | tee $OUT/r4ag_store_nt.txt
