#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $OUT/r3s_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r3s_pytest.log | tail -2
timeout 900 bash tools/profile_gpu.sh r3final2 > $OUT/r3s_profile.log 2>&1; echo "profile rc=$?"; tail -6 $OUT/r3s_profile.log
cp $OUT/profile_summary_r3final2.json $ROOT/profiles/r3_profile_summary.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r3s_bench.json 2>$OUT/r3s_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/r3s_bench.json")); t=d["train"]
print("sampler", d["ms_per_step"], d["value"], "train", t["value"], "pmc_stale", d["config"].get("pmc_stale"), "frac", d["roofline"]["frac"])
print({k:v.get("value") for k,v in d["extra"].items()})
PY
timeout 300 python tools/full_sampler_run.py 256 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/r3s_full_sampler_run.txt
