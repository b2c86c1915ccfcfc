#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (rocprofv3 --pmc, --kernel-trace only) over one convolution shape on the
# F(4x4,3x3) kernel (tile 6) and on the F(2x2,3x3) kernel (tile 5); prints per-kernel averages
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/w4_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPE=${SHAPE:-"128 128 32 1"}
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
for T in 6 5; do
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    rocprofv3 --pmc $P --kernel-trace -d $OUT/t${T}_p$i -o c --output-format csv -- python $ROOT/tools/conv_one.py $T $SHAPE > $OUT/t${T}_p$i.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
for T in (6, 5):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/t%d_p*/**/*counter_collection.csv" % T, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wino" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== tile", T, "(per launch averages)")
    for k in sorted(agg):
        print("  %-28s %.4g" % (k, sum(agg[k]) / len(agg[k])))
PY
