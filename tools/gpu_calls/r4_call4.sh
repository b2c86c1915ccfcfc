#!/bin/bash
# round 4, call 4: upper bounds for conv_wino4_kernel (no stage barrier / no epilogue store: timing only), power draw during the
# sampler, the one-launch column sums in the training step, the N > 1 code paths on one GPU over gloo, per-op dump
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
{ for i in 1 2; do
  timeout 120 python tools/w4_bounds.py 256
  for v in w4nobar w4nostore w4nobarnostore; do SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_$v.so timeout 120 python tools/w4_bounds.py 256; done
done; } 2>&1 | grep -v amdgpu.ids > $OUT/r4d_w4_bounds.txt; cat $OUT/r4d_w4_bounds.txt
# power / clocks while the sampler runs (rocm-smi polled beside bench.py)
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done ) > $OUT/r4d_power.txt 2>&1 &
SMI=$!
timeout 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras --no-train --dump-ops $OUT/r4d_ops.json > $OUT/r4d_bench.json 2> $OUT/r4d_bench.err
wait $SMI
python - <<PY
import json, re
rows = [l for l in open("$OUT/r4d_power.txt") if l.startswith("{")]
pw, ck = [], []
for l in rows:
    try: d = json.loads(l)
    except Exception: continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "ower" in k and "W" in k:
            try: pw.append(float(v))
            except Exception: pass
        if "sclk" in k:
            m = re.search(r"(\d+)Mhz", str(v)); 
            if m: ck.append(int(m.group(1)))
print("power W samples:", pw)
print("sclk MHz samples:", ck)
if rows: print("keys:", list(json.loads(rows[0]).get("card0", {}).keys()))
PY
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "backward_kernels or whole or fused" > $OUT/r4d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4d_pytest.log
for f in 0 1 0 1; do
  SSDE_COLSUM_FUSED=$f timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r4d_train_$f.json 2> $OUT/r4d_train_$f.err
  python - <<PY
import json
l = [x for x in open("$OUT/r4d_train_$f.json") if x.startswith("{")]
d = json.loads(l[-1])["train"]
print("colsum fused=$f", round(d["value"], 5), {k: round(v["ms"], 3) for k, v in d.get("by_class", {}).items()})
PY
done
timeout 900 python bench.py --gpus 2 --dist-backend gloo --share-device --steps 5 --warmup 2 --no-cpu-baseline --no-extras --train-steps 10 --train-warmup 3 > $OUT/r4d_bench_2ranks_gloo_shared.json 2> $OUT/r4d_bench_2ranks.err; echo "2-rank rc=$?"; tail -c 1500 $OUT/r4d_bench_2ranks_gloo_shared.json; tail -5 $OUT/r4d_bench_2ranks.err
