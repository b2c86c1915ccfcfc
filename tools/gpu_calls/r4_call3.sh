#!/bin/bash
# round 4, call 3: the persistent pipelined GEMM (fp32 and bf16x6): parity, A/B against the plain kernels, the sampler with it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "conv1x1 or linear or groupnorm_statistics" > $OUT/r4c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4c_pytest.log
timeout 300 python tools/matrix_ab.py 256 > $OUT/r4c_matrix_ab.txt 2>&1; cat $OUT/r4c_matrix_ab.txt
for v in "f32 0" "f32 1" "bf16x6 1" "f32 0" "f32 1"; do
  set -- $v
  SSDE_MATRIX=$1 SSDE_GEMM_PIPE=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-train > $OUT/r4c_bench_$1_$2.json 2> $OUT/r4c_bench_$1_$2.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4c_bench_$1_$2.json") if x.startswith("{")]
d = json.loads(l[-1])
print("$1 pipe=$2", round(d["value"], 3), round(d["ms_per_step"], 2), {k: round(v.get("ms"), 3) for k, v in d["roofline"]["by_class"].items()})
PY
done
SSDE_WINOGRAD=1 timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q > $OUT/r4c_pytest_full.log 2>&1; echo "pytest full rc=$?"; tail -5 $OUT/r4c_pytest_full.log
