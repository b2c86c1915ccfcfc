#!/bin/bash
# round 5, call 7: the whole GPU suite on the ABI-8 sources; training step with / without the batched finishing launches and the step graph
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for CFG in "0 0" "1 0" "1 1" "0 0" "1 1"; do
  set -- $CFG
  SSDE_DEFER_FINISH=$1 SSDE_TRAIN_GRAPH=$2 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --train-steps 40 --train-warmup 8 > $OUT/r5f_train_$1_$2.json 2> $OUT/r5f_train_$1_$2.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5f_train_$1_$2.json") if x.startswith("{")]
t = json.loads(l[-1])["train"]
print("defer=$1 graph=$2 train", round(t["value"], 5), {k: (round(v["ms"], 3), v.get("launches")) for k, v in t.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r5f_train_ab.txt
