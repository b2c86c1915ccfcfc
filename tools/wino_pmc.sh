#!/bin/bash
# PMC passes over a single Winograd conv shape (runs on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wino_pmc
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys; sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tools")
import conv_bench as cb
from score_sde_pytorch_amd import _lib as L
print(cb.time_conv(256, 256, 256, 16, L.TILE_WINOGRAD, False, reps=3))
print(cb.time_conv(256, 256, 256, 16, L.TILE_AUTO, False, reps=3))
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/p1 -o w --output-format csv -- python /tmp/one.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace -d $OUT/p2 -o w --output-format csv -- python /tmp/one.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv_" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in cs.items(): print("   %-28s %.4g (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
tail -3 $OUT/p1.log $OUT/p2.log
