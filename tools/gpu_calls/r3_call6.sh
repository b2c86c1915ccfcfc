#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_plan_c_host.py -m gpu -x -q > $OUT/r3f_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3f_pytest.log
timeout 1200 bash tools/profile_gpu.sh r3f > $OUT/r3f_profile.log 2>&1; tail -5 $OUT/r3f_profile.log
python - <<PY
import json
d=json.load(open("$OUT/profile_summary_r3f.json"))
for r in sorted(d["kernel_stats"], key=lambda r:-r["total_us"])[:40]:
    print("%-60s calls %5d avg %9.1f us total %10.1f us %5.1f%%" % (r["kernel"][:60], r["calls"], r["avg_us"], r["total_us"], r["percent"]))
print(json.dumps(d.get("mfma",{}).get("conv_wino4_kernel")), json.dumps(d.get("hbm_traffic",{}).get("conv_wino4_kernel")))
PY
