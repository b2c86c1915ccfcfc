#!/bin/bash
# round 5, call 20: the default bench line of the final sources (what the driver runs), then rocprofv3 kernel-trace stats + the PMC
# passes of tools/profile_gpu.sh (FETCH_SIZE, WRITE_SIZE, MFMA busy: one counter group per pass, --kernel-trace only); cycle trace of the two shares of a split launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/r5s_bench_final.json 2> $OUT/r5s_bench_final.err; echo "bench rc=$?"
tail -4 $OUT/r5s_bench_final.err
python - <<PY
import json
l = [x for x in open("$OUT/r5s_bench_final.json") if x.startswith("{")]
d = json.loads(l[-1])
print(d["dtype"][:40], round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms | frac", round(d["roofline"]["frac"], 3), {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
print("   train", round(d["train"]["value"], 5), "| f32:", round(d["matrix_f32"]["sampler"]["value"], 4), d["matrix_f32"].get("train", {}).get("value"))
print("   extra", {k: (round(v.get("value", 0), 5), v.get("nfe")) for k, v in d.get("extra", {}).items()}, "| cpu", d.get("cpu_baseline", {}).get("value"))
PY
timeout 900 bash tools/profile_gpu.sh r5final 2>&1 | tail -25
SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_w4rtrace.so timeout 200 python tools/wino4r_split_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r5s_wino4r_split_trace.txt
