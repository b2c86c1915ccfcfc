#!/bin/bash
# round 5, call 23: the F(4x4,3x3) weight gradient's split sum inside its G^T . G kernel (one launch instead of two): parity on the
# GPU (bit-identical to the two-pass form), training step A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/r5v_pytest.txt
for FR in 0 1 0 1; do
  SSDE_WGRAD4_FUSED_REDUCE=$FR timeout 300 python bench.py --train-only --train-steps 40 --train-warmup 8 > $OUT/r5v_bench.json 2> $OUT/r5v_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5v_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print("SSDE_WGRAD4_FUSED_REDUCE=$FR train", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d["by_class"].items()})
PY
done 2>&1 | tee $OUT/r5v_wgrad4_fused_reduce_ab.txt
