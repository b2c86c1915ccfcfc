#!/bin/bash
# round 4, call 2: shapes of the bf16x6 GEMM (rows per workgroup x load-ahead depth), its cycle trace, and an A-B-A-B of the
# whole sampler (is the class shift of call 1 DVFS or run-to-run noise?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python tools/matrix_ab.py 256 > $OUT/r4b_matrix_ab.txt 2>&1; cat $OUT/r4b_matrix_ab.txt
for v in "128 1" "64 1" "64 2"; do
  set -- $v
  echo "== trace x6 BM=$1 PF=$2"
  SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_gemmtrace.so SSDE_MATRIX=bf16x6 SSDE_X6_BM=$1 SSDE_X6_PF=$2 timeout 120 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
done > $OUT/r4b_gemm_trace.txt 2>&1; cat $OUT/r4b_gemm_trace.txt
for m in f32 bf16x6 f32 bf16x6; do
  SSDE_MATRIX=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-train > $OUT/r4b_bench_$m.json 2> $OUT/r4b_bench_$m.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4b_bench_$m.json") if x.startswith("{")]
d = json.loads(l[-1])
print("$m", round(d["value"], 3), round(d["ms_per_step"], 2), {k: round(v.get("ms"), 3) for k, v in d["roofline"]["by_class"].items()})
PY
done
