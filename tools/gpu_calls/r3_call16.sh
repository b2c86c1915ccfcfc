#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/ -m gpu -x -q > $OUT/r3p_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3p_pytest.log
timeout 900 python bench.py > $OUT/r3p_bench.json 2>$OUT/r3p_bench.err; echo "bench rc=$?"; tail -8 $OUT/r3p_bench.err
python - <<PY
import json
d=json.load(open("$OUT/r3p_bench.json")); t=d["train"]
print("sampler", d["ms_per_step"], d["value"], "train", t["value"])
print({k:v.get("value") for k,v in d["extra"].items()})
print(d["roofline"].get("sustained_mfma_probe"))
PY
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4trace.so timeout 200 python tools/wino4_trace.py 2>&1 | grep -v amdgpu > $OUT/r3p_w4_trace.txt
grep "==\|clock\|total" $OUT/r3p_w4_trace.txt | head -20
timeout 900 bash tools/profile_gpu.sh r3final > $OUT/r3p_profile.log 2>&1; echo "profile rc=$?"; tail -5 $OUT/r3p_profile.log
timeout 600 python bench.py --workload subvp_likelihood --likelihood-tol 1e-5 > $OUT/r3p_bench_likelihood.json 2>/dev/null; echo "likelihood rc=$?"; head -c 900 $OUT/r3p_bench_likelihood.json
