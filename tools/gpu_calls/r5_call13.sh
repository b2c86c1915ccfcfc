#!/bin/bash
# round 5, call 13: the two-kernel rule after the dealing change (SSDE_WINO4_TWO=3: four cout tiles up; default: + input <= 2 x output; 2: everywhere)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for TWO in 3 1 2 3 1; do
  SSDE_WINO4_TWO=$TWO timeout 200 python bench.py --matrix f32 --no-other-matrix --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-train > $OUT/r5l_bench.json 2> $OUT/r5l_bench.err
  python - <<PY
import json
l = [x for x in open("$OUT/r5l_bench.json") if x.startswith("{")]
d = json.loads(l[-1])
print("two=$TWO", round(d["value"], 4), "img/s", round(d["ms_per_step"], 2), "ms |", {k: round(v["ms"], 3) for k, v in d["roofline"]["by_class"].items()})
PY
done 2>&1 | tee $OUT/r5l_two_kernel_rule_ab.txt
timeout 200 python bench.py --workload ffhq256 --matrix f32 --no-other-matrix --steps 5 --warmup 2 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ffhq256 default rule', round(d['value'],5), round(d['ms_per_step'],2))" | tee -a $OUT/r5l_two_kernel_rule_ab.txt
SSDE_WINO4_TWO=3 timeout 200 python bench.py --workload ffhq256 --matrix f32 --no-other-matrix --steps 5 --warmup 2 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ffhq256 two=3', round(d['value'],5), round(d['ms_per_step'],2))" | tee -a $OUT/r5l_two_kernel_rule_ab.txt
