#!/bin/bash
# round 4, call 34: smoke() and the default bench line (as the driver runs it, N=1) on the final sources
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python bench.py > $OUT/bench_r4_final2.json 2> $OUT/bench_r4_final2.err ) 2>&1 | tail -3
python - <<PY
import json
l = [x for x in open("$OUT/bench_r4_final2.json") if x.startswith("{")]
d = json.loads(l[-1])
print(d["metric"], d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "train", d["train"]["value"])
print({k: (round(v["ms"], 3), round(v.get("frac", 0), 3)) for k, v in d["roofline"]["by_class"].items()})
print({k: round(v["ms"], 3) for k, v in d["train"]["by_class"].items()})
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("extra", {}).items()})
PY
