#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (one counter group per pass, --kernel-trace only) over the kernel micro-benchmarks,
# to see what the matrix kernels wait on.  Usage: tools/pmc_kernels.sh <tag>  -> gpurun_out/pmc_<tag>/<group>.csv
set -u
TAG=${1:-k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/pmc_workload.py"
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace -d $OUT/g$i -o k --output-format csv -- $CMD > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("conv_wino_kernel<true>", "conv_wino_kernel<false>", "gemm1x1_kernel<false>", "wgrad_wino_kernel<false>", "wgrad_kernel<1", "conv_mfma_kernel"):
            if key in k:
                a = res[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} for k, cs in res.items()}
json.dump(out, open("$ROOT/gpurun_out/pmc_$TAG.json", "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s %16.0f" % (c, v))
PY
