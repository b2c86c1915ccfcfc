#!/bin/bash
# round 4, call 8: the forward by-product in the coalesced [pos][C/4][t][4] layout: parity, A/B of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "fed_by_the_forward or whole_network or fused_training" > $OUT/r4h_pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -3 $OUT/r4h_pytest_train.log
for f in 0 1 0 1; do
  SSDE_WINO_V_FROM_FORWARD=$f timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4h_train_$f.json 2> $OUT/r4h_train_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4h_train_$f.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("V from forward=$f", round(d["value"], 5), "arena GB", round(d["arena_gb"], 2), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done
timeout 900 python -m pytest tests/test_bench_sizes_gpu.py -m gpu -x -q -k "gradients" > $OUT/r4h_pytest_sizes.log 2>&1; echo "pytest sizes rc=$?"; tail -3 $OUT/r4h_pytest_sizes.log
