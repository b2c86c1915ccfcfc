#!/bin/bash
# round 4, call 1: the bf16x6 experiment -- parity of the split GEMM, its A/B, and the sampler in both modes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv1x1 or linear" > $OUT/r4a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4a_pytest.log
timeout 300 python tools/matrix_ab.py 256 > $OUT/r4a_matrix_ab.txt 2>&1; cat $OUT/r4a_matrix_ab.txt
for m in f32 bf16x6; do
  SSDE_MATRIX=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-train > $OUT/r4a_bench_$m.json 2> $OUT/r4a_bench_$m.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4a_bench_$m.json") if x.startswith("{")]
d = json.loads(l[-1])
print("$m", d["value"], d["ms_per_step"], {k: (v.get("ms"), v.get("frac")) for k, v in d["roofline"]["by_class"].items()})
PY
done
SSDE_MATRIX=bf16x6 timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -m gpu -x -q > $OUT/r4a_pytest_x6.log 2>&1; echo "pytest x6 rc=$?"; tail -3 $OUT/r4a_pytest_x6.log
