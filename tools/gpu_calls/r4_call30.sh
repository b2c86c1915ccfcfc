#!/bin/bash
# round 4, call 30: two-kernel F(4x4,3x3) everywhere (SSDE_WINO4_TWO=2) against the rule (from four cout tiles up), sampler and training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for f in 1 2 1 2; do
  SSDE_WINO4_TWO=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4ac_bench_$f.json 2> $OUT/r4ac_bench_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4ac_bench_$f.json") if x.startswith("{")]
d = json.loads(l[-1])
print("SSDE_WINO4_TWO=$f", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"].get("by_class", {}).items()})
t = d["train"]
print("   train", round(t["value"], 5), {k: round(v["ms"], 3) for k, v in t.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r4ac_two_kernels_everywhere.txt
