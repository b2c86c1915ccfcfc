#!/bin/bash
# round 5, call 5: conv_wino4r with a two-stage-deep register ring (two whole stages of loads in flight per wave), 8-wave and
# 2 x 4-wave workgroups, with and without the half-tile start offset of the second workgroup of a CU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "two_kernels or register_fed or repack or dropout" 2>&1 | tail -4
{
W4_BOUNDS_TWO=1 timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids
for ST in 1 3 0; do echo "== SSDE_DEBUG=w4r_stagger=$ST"; SSDE_DEBUG=w4r_stagger=$ST W4_BOUNDS_TWO=r timeout 200 python tools/w4_bounds.py 256 2>&1 | grep -v amdgpu.ids; done
for ST in 0 1; do echo "== trace, 2 x 4 waves, stagger $ST"; SSDE_DEBUG=w4r_stagger=$ST SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids; done
echo "== trace, 8 waves"; SSDE_W4R_WIDE=1 SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 120 python tools/wino4r_trace.py 2>&1 | grep -v amdgpu.ids
} > $OUT/r5d_wino4r_deep.txt 2>&1
cat $OUT/r5d_wino4r_deep.txt
for CFG in "0 0 1" "1 0 1" "0 1 1" "0 0 2" "0 1 2"; do
  set -- $CFG
  SSDE_W4R_WIDE=$1 SSDE_DEBUG=w4r_stagger=$2 SSDE_WINO4_TWO=$3 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --train-steps 30 --train-warmup 5 > $OUT/r5d_bench_$1_$2_$3.json 2> $OUT/r5d_bench_$1_$2_$3.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5d_bench_$1_$2_$3.json") if x.startswith("{")]
d = json.loads(l[-1])
print("wide=$1 stagger=$2 two=$3", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
t = d["train"]
print("   train", round(t["value"], 5), {k: round(v["ms"], 3) for k, v in t.get("by_class", {}).items()})
PY
done 2>&1 | tee $OUT/r5d_bench_ab.txt
