#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
(for S in 0 3000 6000 12000 24000 48000; do
  echo "== SSDE_GEMM_STAGGER=$S"
  SSDE_GEMM_STAGGER=$S timeout 200 python tools/gemm_bench.py 256 2>&1 | grep -v amdgpu | head -7
done) | tee $OUT/r3u_gemm_stagger.txt
for S in 0 6000 12000 24000; do SSDE_GEMM_STAGGER=$S timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train-steps 10 > $OUT/r3u_bench_s$S.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$OUT/r3u_bench_s$S.json")); t=d["train"]
print("SSDE_GEMM_STAGGER=$S sampler ms", d["ms_per_step"], "gemm class ms", d["roofline"]["by_class"]["conv1x1_gemm"]["ms"], "train s/step", t["value"])
PY
done
