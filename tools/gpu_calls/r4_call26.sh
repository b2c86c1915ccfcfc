#!/bin/bash
# round 4, call 26: two-kernel F(4x4,3x3) in the networks: GPU parity, then A-B-A-B of the sampler and of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -x -q -k "two_kernels or golden" > $OUT/r4y_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4y_pytest.log
for f in 0 1 0 1; do
  SSDE_WINO4_TWO=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4y_bench_$f.json 2> $OUT/r4y_bench_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4y_bench_$f.json") if x.startswith("{")]
d = json.loads(l[-1])
print("two kernels=$f", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"].get("by_class", {}).items()})
t = d["train"]
print("   train", round(t["value"], 5), {k: round(v["ms"], 3) for k, v in t.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r4y_two_kernels_e2e.txt
