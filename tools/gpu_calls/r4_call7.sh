#!/bin/bash
# round 4, call 7: the F(4x4,3x3) weight gradient fed by the forward launch's by-product: parity on the GPU, A/B of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -x -q > $OUT/r4g_pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -3 $OUT/r4g_pytest_train.log
for f in 0 1 0 1; do
  SSDE_WINO_V_FROM_FORWARD=$f timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4g_train_$f.json 2> $OUT/r4g_train_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4g_train_$f.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("V from forward=$f", round(d["value"], 5), "arena GB", round(d["arena_gb"], 2), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done
timeout 900 python -m pytest tests/test_bench_sizes_gpu.py -m gpu -x -q -k "gradients" > $OUT/r4g_pytest_sizes.log 2>&1; echo "pytest sizes rc=$?"; tail -3 $OUT/r4g_pytest_sizes.log
