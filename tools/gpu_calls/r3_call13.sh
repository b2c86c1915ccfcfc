#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py tests/test_sampler_gpu.py -m gpu -x -q -k "winograd or generic_loss or split or trajectory or golden" > $OUT/r3m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3m_pytest.log
(for B in 256 128 64; do
  for KS in 1 0; do
    echo "== B=$B SSDE_CONV_KSPLIT=$KS"
    CONV_BENCH_RESID=1 CONV_BENCH_SHAPES="256,256,8;512,256,8;384,256,8;256,256,16;512,256,16" SSDE_CONV_KSPLIT=$KS timeout 300 python tools/conv_bench.py $B 2>&1 | grep -v amdgpu | grep "gn=1"
  done
done) | tee $OUT/r3m_conv_split.txt
for MIN in 192 1000000; do SSDE_W4_SPLIT_MIN_WGS=$MIN timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --train-steps 10 > $OUT/r3m_bench_min$MIN.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/r3m_bench_min$MIN.json")); t=d["train"]
print("SSDE_W4_SPLIT_MIN_WGS=$MIN sampler ms/iter", d["ms_per_step"], "img/s", d["value"], "train s/step", t["value"], {k:round(v["ms"],2) for k,v in t.get("by_class",{}).items()})
PY
done
