#!/bin/bash
# round 4, call 14: conv_wino4x (F(4x4,3x3) on the BF16 matrix pipe): parity on the GPU, timing against conv_wino4, the sampler
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "bf16_matrix_pipe" > $OUT/r4n_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r4n_pytest.log
W4_BOUNDS_SPLIT=1 timeout 300 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > $OUT/r4n_w4x.txt; cat $OUT/r4n_w4x.txt
for m in f32 bf16x6 f32 bf16x6; do
  SSDE_MATRIX=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-train --no-roofline > $OUT/r4n_bench_$m.json 2> $OUT/r4n_bench_$m.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4n_bench_$m.json") if x.startswith("{")]
if l:
    d = json.loads(l[-1]); print("$m", round(d["value"], 3), round(d["ms_per_step"], 2), d["config"]["state_finite"])
else:
    print("$m failed"); print(open("$OUT/r4n_bench_$m.err").read()[-1500:])
PY
done
SSDE_MATRIX=bf16x6 timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q > $OUT/r4n_pytest_x6.log 2>&1; echo "pytest x6 rc=$?"; tail -6 $OUT/r4n_pytest_x6.log
