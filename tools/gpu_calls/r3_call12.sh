#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "wgrad or gradients or fused_step or backward" > $OUT/r3l_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3l_pytest.log
(echo "== F(4x4,3x3) wherever legal"; SSDE_WGRAD_WINOGRAD=44 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"
echo "== F(2x2,3x3)"; SSDE_WGRAD_WINOGRAD=2 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2") | tee $OUT/r3l_wgrad_bench.txt
for M in 4 2; do SSDE_WGRAD_WINOGRAD=$M timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --train-steps 10 > $OUT/r3l_bench_wg$M.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/r3l_bench_wg$M.json")); t=d["train"]
print("SSDE_WGRAD_WINOGRAD=$M train s/step", t["value"], "sampler ms", d["ms_per_step"], {k:round(v["ms"],2) for k,v in t.get("by_class",{}).items()})
PY
done
