#!/bin/bash
# round 5, call 12: conv_wino4r with every V run loaded once per workgroup (wave = 4 positions x both cout halves + half a position:
# 56 KB per stage and CU instead of 72) against the previous dealing (libssde_hip_w4rnobal.so: timing only -- it reads the new weight
# image in its old order): parity, per layer, cycle trace, sampler + training step A-B-A-B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels or register_fed or repack" 2>&1 | tail -2
{
for LIB in "" $ROOT/tools/variants/libssde_hip_w4rnobal.so "" $ROOT/tools/variants/libssde_hip_w4rnobal.so; do
  SSDE_LIB_PATH=$LIB W4_BOUNDS_TWO=1 timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids
done
W4R_TRACE_BID=300 SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids
} > $OUT/r5k_wino4r_dedup.txt 2>&1
grep -v "epilogue rounds" $OUT/r5k_wino4r_dedup.txt
for LIB in "" $ROOT/tools/variants/libssde_hip_w4rnobal.so "" $ROOT/tools/variants/libssde_hip_w4rnobal.so; do
  SSDE_LIB_PATH=$LIB timeout 200 python bench.py --matrix f32 --no-other-matrix --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r5k_bench.json 2> $OUT/r5k_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5k_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print("lib=$(basename "$LIB")", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()}, "| train", round(d["train"]["value"], 5))
PY
done 2>&1 | tee $OUT/r5k_bench_ab.txt
