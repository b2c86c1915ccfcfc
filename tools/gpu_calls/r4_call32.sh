#!/bin/bash
# round 4, call 32: the default bench line (as the driver runs it, N=1) on the final sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
( time timeout 1500 python bench.py > $OUT/bench_r4_final.json 2> $OUT/bench_r4_final.err ) 2>&1 | tail -3
tail -c 1500 $OUT/bench_r4_final.err
python - <<PY
import json
l = [x for x in open("$OUT/bench_r4_final.json") if x.startswith("{")]
d = json.loads(l[-1])
print(d["metric"], d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "train", d["train"]["value"])
print({k: (round(v["ms"], 3), round(v.get("frac", 0), 3)) for k, v in d["roofline"]["by_class"].items()})
print("cpu", d.get("cpu_baseline"))
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("extra", {}).items()})
PY
