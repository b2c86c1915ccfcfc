#!/bin/bash
# round 5, call 28: a short bench line of the final sources (sampler + roofline + training step; no CPU baseline, no other configs,
# no other matrix mode) with the corrected FIR byte count
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-matrix --train-steps 40 --train-warmup 8 > $OUT/r5y_bench_short.json 2> $OUT/r5y_bench_short.err; echo "rc=$?"
python - <<PY
import json
l = [x for x in open("$OUT/r5y_bench_short.json") if x.startswith("{")]
d = json.loads(l[-1])
r = d["roofline"]
print(round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms | frac", round(r["frac"], 3), "| train", round(d["train"]["value"], 5), "| pmc_stale", d["config"].get("pmc_stale"))
print({k: (round(v["ms"], 3), v.get("launches"), round(v.get("frac", 0), 3)) for k, v in r["by_class"].items()})
print({k: (round(v["ms"], 3), v.get("launches"), round(v.get("frac", 0), 3)) for k, v in d["train"]["by_class"].items()})
PY
