#!/bin/bash
# round 4, call 9: the whole GPU suite and the driver's own bench command (default flags), timed
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r4i_pytest_gpu.log 2>&1; echo "pytest gpu rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $OUT/r4i_pytest_gpu.log
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r4i_bench_default.json 2> $OUT/r4i_bench_default.err; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
grep "^\[bench" $OUT/r4i_bench_default.err | tail -12
python - <<PY
import json
l = [x for x in open("$OUT/r4i_bench_default.json") if x.startswith("{")]
d = json.loads(l[-1])
print("value", d["value"], "ms", d["ms_per_step"], "train", d["train"]["value"], "frac", d["roofline"]["frac"])
print("extra keys", list(d["extra"].keys()))
print("x6", d["extra"]["matrix_bf16x6"])
print("lik", {k: d["extra"]["subvp_likelihood"][k] for k in ("value", "nfe", "seconds_per_solve", "rtol_atol")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:120], d["cpu_baseline"]["train"]["value"], d["cpu_baseline"]["train"]["sample"][:100])
PY
