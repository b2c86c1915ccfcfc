#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "wgrad or gradients or backward or fused_step" > $OUT/r3v_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r3v_pytest.log | tail -1
(for CFG in "2 1" "1 1" "2 0" "2 1"; do set -- $CFG
  echo "== SSDE_WGRAD4_XVEC=$1 SSDE_WGRAD4_XCD=$2"
  SSDE_WGRAD_WINOGRAD=44 SSDE_WGRAD4_XVEC=$1 SSDE_WGRAD4_XCD=$2 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"
done) | tee $OUT/r3v_wgrad4_xvec_xcd.txt
for CFG in "2 1" "1 0"; do set -- $CFG
SSDE_WGRAD4_XVEC=$1 SSDE_WGRAD4_XCD=$2 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --train-steps 10 > $OUT/r3v_bench_$1$2.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/r3v_bench_$1$2.json")); t=d["train"]
print("XVEC=$1 XCD=$2 train s/step", t["value"], {k:round(v["ms"],2) for k,v in t.get("by_class",{}).items()})
PY
done
