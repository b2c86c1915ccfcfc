#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/ -m gpu -x -q > $OUT/r3w_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r3w_pytest.log | tail -2
timeout 900 bash tools/profile_gpu.sh r3final3 > $OUT/r3w_profile.log 2>&1; echo "profile rc=$?"; tail -3 $OUT/r3w_profile.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r3w_bench.json 2>$OUT/r3w_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/r3w_bench.json")); t=d["train"]
print("sampler", d["ms_per_step"], d["value"], "train", t["value"], "frac", d["roofline"]["frac"], {k:round(v["ms"],2) for k,v in t["by_class"].items()})
print({k:v.get("value") for k,v in d["extra"].items()})
PY
