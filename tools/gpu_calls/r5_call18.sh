#!/bin/bash
# round 5, call 18: eight shares on the 4x4 maps (layers, parity, sampler A-B-A-B with SSDE_W4R_SPLIT8), conv_small with batched staging loads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/w4r_split4x4_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/r5q_w4r_split_4x4_layers.txt
timeout 200 python tools/conv_small_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r5q_conv_small_layers.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "image_channels or register_fed or two_kernels" 2>&1 | tail -3 | tee $OUT/r5q_pytest.txt
for S8 in 0 1 0 1; do
  SSDE_W4R_SPLIT8=$S8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-matrix --no-train > $OUT/r5q_bench.json 2> $OUT/r5q_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5q_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print("SSDE_W4R_SPLIT8=$S8", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
PY
done 2>&1 | tee $OUT/r5q_split8_bench_ab.txt
