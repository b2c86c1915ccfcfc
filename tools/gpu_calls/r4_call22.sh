#!/bin/bash
# round 4, call 22: the one-pass GroupNorm backward with 1024-thread (one per CU) and 512-thread (two per CU) workgroups
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SSDE_GN_BWD_THREADS=512 timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "backward_kernels or whole_network_gradients_small" 2>&1 | tail -2
for f in 1024 512 1024 512; do
  SSDE_GN_BWD_THREADS=$f timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r4u_train_$f.json 2> $OUT/r4u_train_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4u_train_$f.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("threads=$f", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r4u_gn_bwd_threads_ab.txt
