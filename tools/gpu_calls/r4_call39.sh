#!/bin/bash
# round 4, call 39: stream-K split of the F(4x4,3x3) weight-gradient GEMM (SSDE_WGRAD4_STREAMK=1): parity, per-layer and training-step A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 120 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "stream_k" 2>&1 | tail -2
{
for SK in 0 1; do
  echo "== SSDE_WGRAD4_STREAMK=$SK"
  SSDE_WGRAD4_STREAMK=$SK timeout 100 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu.ids | grep "pro=2"
done
for SK in 0 1; do
  SSDE_WGRAD4_STREAMK=$SK timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --train-steps 20 --train-warmup 4 > $OUT/r4ah_train_$SK.json 2> $OUT/r4ah_train_$SK.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4ah_train_$SK.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("stream-K=$SK", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done
} 2>&1 | tee $OUT/r4ah_wgrad4_streamk.txt
