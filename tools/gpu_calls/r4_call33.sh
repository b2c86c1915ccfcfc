#!/bin/bash
# round 4, call 33: per-kernel times of the F(4x4,3x3) weight gradient at the training shapes (rocprofv3 --kernel-trace), one shape per process
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for S in "32,128,0,128" "32,256,0,128" "16,256,0,256" "16,256,256,256" "32,384,0,128"; do
  rm -rf $OUT/r4ad_tr
  rocprofv3 --kernel-trace --stats -d $OUT/r4ad_tr -o t --output-format csv -- python - <<PY > /dev/null 2>&1
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tools")
import wgrad_bench as wb
h, c1, c2, co = [int(v) for v in "$S".split(",")]
wb.time_wgrad(128, h, c1, c2, co, 2, reps=10)
PY
  python - <<PY
import csv, glob
f = glob.glob("$OUT/r4ad_tr/**/*kernel_stats.csv", recursive=True)[0]
h, c1, c2, co = [int(v) for v in "$S".split(",")]
T = 128 * (h // 4) ** 2
gf = 2.0 * 36 * T * (c1 + c2) * co / 1e9
rows = [r for r in csv.DictReader(open(f)) if "wgrad4" in r["Name"] or "wino4_xform" in r["Name"] or "wgrad_wino" in r["Name"]]
print("shape %dx%d cin=%d+%d cout=%d: GEMM work %.2f GFLOP" % (h, h, c1, c2, co, gf))
for r in rows:
    us = float(r["AverageNs"]) / 1e3
    name = r["Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
    extra = "  -> %.1f TF/s executed = %.2f of 157.3" % (gf / us / 1e3 * 1e3 / 1e3 * 1e3, gf / us / 1e3 * 1e3 / 157.3) if "gemm" in name else ""
    print("   %-40s calls %3s avg %8.1f us%s" % (name, r["Calls"], us, extra))
PY
done 2>&1 | tee $OUT/r4ad_wgrad4_kernels.txt
rm -rf $OUT/r4ad_tr
