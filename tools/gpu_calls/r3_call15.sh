#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "1x1 or backward or gradients or fused_step or wgrad" > $OUT/r3o_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3o_pytest.log
(echo "== pipelined (wgrad1x1_gemm_kernel), library split"; WG_1X1=1 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep "^1x1"
echo "== pipelined, explicit splits"; WG_1X1=1 WG_SPLITS_1X1=32,64,128,256 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep "^1x1" | grep "pro=0"
echo "== chunked (wgrad_kernel<1,2,2>)"; SSDE_WGRAD_1X1_PIPELINED=0 WG_1X1=1 timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep "^1x1") | tee $OUT/r3o_wgrad1x1.txt
for M in 1 0; do SSDE_WGRAD_1X1_PIPELINED=$M timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --train-steps 10 > $OUT/r3o_bench_p$M.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/r3o_bench_p$M.json")); t=d["train"]
print("SSDE_WGRAD_1X1_PIPELINED=$M train s/step", t["value"], "sampler ms", d["ms_per_step"], {k:round(v["ms"],2) for k,v in t.get("by_class",{}).items()})
print(d["roofline"].get("sustained_mfma_probe"))
PY
done
