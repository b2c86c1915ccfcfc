#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/r3e_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/r3e_pytest.log
cat $OUT/f4x4_trajectory_error_growth.txt 2>/dev/null | awk 'NR%5==1'
timeout 1200 python bench.py --steps 10 --warmup 3 > $OUT/r3e_bench.json 2> $OUT/r3e_bench.err; echo "bench rc=$?"; tail -3 $OUT/r3e_bench.err; python - <<PY
import json
d=json.load(open("$OUT/r3e_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["train"]["value"])
print(json.dumps(d["extra"])[:1500])
PY
