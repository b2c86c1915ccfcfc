#!/bin/bash
# round 4, call 36: split rule of the F(4x4,3x3) weight-gradient GEMM (quantisation-aware) against the old one (SSDE_WGRAD4_WGS=768)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "whole_network or weight_gradient or wgrad" 2>&1 | tail -2
{
for W in old new old new; do
  if [ $W = old ]; then export SSDE_WGRAD4_WGS=768; else unset SSDE_WGRAD4_WGS; fi
  echo "== split rule: $W"
  timeout 200 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu.ids | grep "pro=2"
done
unset SSDE_WGRAD4_WGS
for W in old new old new; do
  if [ $W = old ]; then export SSDE_WGRAD4_WGS=768; else unset SSDE_WGRAD4_WGS; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4af_train_$W.json 2> $OUT/r4af_train_$W.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4af_train_$W.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("split rule $W", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done
} 2>&1 | tee $OUT/r4af_wgrad4_split_rule.txt
