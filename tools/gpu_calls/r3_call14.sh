#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/prof_train
rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o tr --output-format csv -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 10 > $OUT/r3n_train_trace.log 2>&1
echo "rc=$?"
find $OUT/prof_train -name "*kernel_stats.csv" | head -3
F=$(find $OUT/prof_train -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
for r in rows[:45]:
    print("%-110s %6s calls %9.1f us avg %8.2f ms total %5s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
find $OUT/prof_train -type f -size +4M -delete
cd $ROOT
timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $OUT/r3n_bench.json 2>$OUT/r3n_bench.err
python - <<PY
import json
d=json.load(open("$OUT/r3n_bench.json")); t=d["train"]
print("sampler", d["ms_per_step"], d["value"], "train", t["value"])
for k,v in t["by_class"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
for k,v in d["roofline"]["by_class"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
