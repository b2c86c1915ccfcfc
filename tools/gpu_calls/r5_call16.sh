#!/bin/bash
# round 5, call 16: the same with the hand-over as 16-byte agent-scope accesses, signalled per round; the engine takes it only where two shares fill the chip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/w4r_split_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/r5o_w4r_split_layers.txt
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels or register_fed" 2>&1 | tail -3 | tee $OUT/r5o_pytest.txt
for SPLIT in 0 1 0 1; do
  SSDE_W4R_SPLIT=$SPLIT timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-other-matrix --train-steps 30 --train-warmup 5 > $OUT/r5o_bench.json 2> $OUT/r5o_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5o_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print("SSDE_W4R_SPLIT=$SPLIT", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()}, "| train", round(d["train"]["value"], 5))
PY
done 2>&1 | tee $OUT/r5o_w4r_split_bench_ab.txt
