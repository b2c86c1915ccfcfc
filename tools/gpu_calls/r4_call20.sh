#!/bin/bash
# round 4, call 20: GroupNorm backward in one pass over dp and x (gn_bwd_fused_kernel): parity, A/B of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "backward_kernels or whole_network or fused_training or dropout" > $OUT/r4s_pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -3 $OUT/r4s_pytest_train.log
for f in 0 1 0 1; do
  SSDE_GN_BWD_FUSED=$f timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4s_train_$f.json 2> $OUT/r4s_train_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4s_train_$f.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("one-pass GroupNorm backward=$f", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r4s_gn_bwd_fused_ab.txt
timeout 900 python -m pytest tests/test_bench_sizes_gpu.py -m gpu -x -q -k "gradients" > $OUT/r4s_pytest_sizes.log 2>&1; echo "pytest sizes rc=$?"; tail -3 $OUT/r4s_pytest_sizes.log
