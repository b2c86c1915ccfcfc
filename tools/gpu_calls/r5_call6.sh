#!/bin/bash
# round 5, call 6: per-layer PMC passes of the F(4x4,3x3) matrix kernels (where do the bytes go?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5e_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SHAPE in "128 128 32" "256 256 16"; do
  TAG=$(echo $SHAPE | tr ' ' '_')
  for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum"; do
    P=$(echo $PASS | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $PASS --kernel-trace -d $OUT/${TAG}_$P -o l --output-format csv -- python $ROOT/tools/w4r_layer.py $SHAPE > $OUT/${TAG}_$P.log 2>&1
    echo "$TAG $P rc=$?"
  done
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wino4" not in k: continue
            k = k.split("(")[0][-40:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(os.path.basename(d.rstrip("/")), k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
find $OUT -type f -size +4M -delete
