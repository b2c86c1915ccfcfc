#!/bin/bash
# round 5, call 3: the register-fed F(4x4,3x3) matrix kernel (conv_wino4r.hip): parity on the GPU, per-layer A/B against the fused kernel
# and the LDS-fed matrix kernel, cycle trace, and the sampler / training step under the three lowering rules
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels or register_fed or repack or dropout" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -x -q -k "two_kernels or golden" 2>&1 | tail -3
{
for i in 1 2; do W4_BOUNDS_TWO=1 timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids; done
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids
} > $OUT/r5b_wino4r_layers.txt 2>&1
cat $OUT/r5b_wino4r_layers.txt
for CFG in "lds 1" "regs 1" "regs 2" "lds 1" "regs 2"; do
  set -- $CFG
  SSDE_WINO4_FEED=$1 SSDE_WINO4_TWO=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r5b_bench_$1_$2.json 2> $OUT/r5b_bench_$1_$2.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5b_bench_$1_$2.json") if x.startswith("{")]
d = json.loads(l[-1])
print("feed=$1 two=$2", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
t = d["train"]
print("   train", round(t["value"], 5), {k: round(v["ms"], 3) for k, v in t.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r5b_bench_ab.txt
