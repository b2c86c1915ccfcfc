#!/bin/bash
# round 5, call 21: the default bench line again, now that profiles/r5_profile_summary.json (the PMC passes of call 20, same kernel
# sources) is in the tree for bench.py to read roofline.traffic / mfma_busy_pmc from; the FFHQ-256 line with its own roofline object
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/r5t_bench_final_with_pmc.json 2> $OUT/r5t_bench_final_with_pmc.err; echo "bench rc=$?"
tail -2 $OUT/r5t_bench_final_with_pmc.err
python - <<PY
import json
l = [x for x in open("$OUT/r5t_bench_final_with_pmc.json") if x.startswith("{")]
d = json.loads(l[-1])
r = d["roofline"]
print(round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms | frac", round(r["frac"], 3), "traffic", r["traffic"], r.get("traffic_parts"), "alg", r["traffic_algorithmic"], "mfma_busy_pmc", r["mfma_busy_pmc"], d["config"].get("pmc_source"), d["config"].get("pmc_stale"))
print("   train", round(d["train"]["value"], 5), "| f32:", round(d["matrix_f32"]["sampler"]["value"], 4), d["matrix_f32"].get("train", {}).get("value"))
print("   extra", {k: (round(v.get("value", 0), 5), v.get("nfe")) for k, v in d.get("extra", {}).items()})
PY
timeout 300 python bench.py --workload ffhq256 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r5t_bench_ffhq256.json 2> $OUT/r5t_bench_ffhq256.err; echo "ffhq rc=$?"
python - <<PY
import json
l = [x for x in open("$OUT/r5t_bench_ffhq256.json") if x.startswith("{")]
d = json.loads(l[-1])
print("ffhq256", round(d["value"], 5), "img/s", round(d["ms_per_step"], 2), "ms |", {k: (round(v["ms"], 3), v.get("launches"), round(v.get("frac", 0), 3)) for k, v in d.get("roofline", {}).get("by_class", {}).items()})
PY
