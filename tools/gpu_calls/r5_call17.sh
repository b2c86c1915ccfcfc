#!/bin/bash
# round 5, call 17: conv_small.hip (image heads) timing + parity; the 8x8 maps at batch 128 with four shares (training step A-B-A-B)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/conv_small_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r5p_conv_small_layers.txt
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "image_channels or register_fed" 2>&1 | tail -3 | tee $OUT/r5p_pytest.txt
timeout 200 python tools/w4r_split_bench.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/r5p_w4r_split_layers_b128.txt
for S4 in 0 1 0 1; do
  SSDE_W4R_SPLIT4=$S4 timeout 300 python bench.py --train-only --train-steps 40 --train-warmup 8 > $OUT/r5p_bench.json 2> $OUT/r5p_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5p_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
t = d.get("train", d)
print("SSDE_W4R_SPLIT4=$S4 train", round(t["value"], 5), {k: round(v["ms"], 3) for k, v in t["by_class"].items()})
PY
done 2>&1 | tee $OUT/r5p_train_split4_ab.txt
for SM in 0 1; do
  SSDE_CONV_SMALL=$SM timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-matrix --no-train > $OUT/r5p_bench2.json 2> $OUT/r5p_bench2.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5p_bench2.json") if x.startswith("{")]
d = json.loads(l[-1])
print("SSDE_CONV_SMALL=$SM", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
PY
done 2>&1 | tee $OUT/r5p_conv_small_bench_ab.txt
