#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py tests/test_bench_sizes_gpu.py -m gpu -x -q > $OUT/r3q_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r3q_pytest.log | tail -2
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/prof_train2
rocprofv3 --kernel-trace --stats -d $OUT/prof_train2 -o tr --output-format csv -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 10 > $OUT/r3q_train_trace.log 2>&1
F=$(find $OUT/prof_train2 -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
for r in rows:
    if any(k in r["Name"] for k in ("prologue_bwd","gn_bwd","colsum","wgrad1x1","wgrad_reduce")):
        print("%-90s %6s calls %9.1f us avg %8.2f ms total" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $OUT/prof_train2 -type f -size +4M -delete
cd $ROOT
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train-steps 10 > $OUT/r3q_bench.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$OUT/r3q_bench.json")); t=d["train"]
print("sampler", d["ms_per_step"], d["value"], "train", t["value"], {k:round(v["ms"],2) for k,v in t["by_class"].items()})
PY
