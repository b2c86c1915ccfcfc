#!/bin/bash
# round 5, call 8: the whole GPU suite in both matrix modes (conftest.MATRIX_MODES); bench line with the bf16x6 headline and the f32 legs beside it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 40 --train-warmup 8 > $OUT/r5g_bench_short.json 2> $OUT/r5g_bench_short.err
tail -3 $OUT/r5g_bench_short.err
python - <<PY
import json
l = [x for x in open("$OUT/r5g_bench_short.json") if x.startswith("{")]
d = json.loads(l[-1])
print(d["dtype"][:40], round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
t = d["train"]
print("   train", round(t["value"], 5), {k: (round(v["ms"], 3), v.get("launches")) for k, v in t.get("by_class", {}).items()})
m = d.get("matrix_f32", {})
print("   f32:", {k: (v if not isinstance(v, dict) else {a: (round(b, 5) if isinstance(b, float) else b) for a, b in v.items()}) for k, v in m.items() if k != "dtype"})
PY
timeout 200 python bench.py --train-only --train-steps 20 --train-warmup 5 --no-roofline | cut -c1-600
