#!/bin/bash
# round 4, call 21: kernel-level times of the GroupNorm backward, one-pass kernel against the three kernels (rocprofv3 --kernel-trace --stats)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  SSDE_GN_BWD_FUSED=$f rocprofv3 --kernel-trace --stats -d $OUT/r4t_trace_$f -o t --output-format csv -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 4 --train-warmup 1 > $OUT/r4t_$f.log 2>&1
  echo "== SSDE_GN_BWD_FUSED=$f"
  python - <<PY
import csv, glob
f = glob.glob("$OUT/r4t_trace_$f/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("gn_bwd", "prologue_bwd", "colsum", "gn_part_finalize")):
        print("%-60s calls %5s avg %9.1f us total %8.2f ms" % (n.split("(")[0][-60:] if not n.startswith("void") else n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find $OUT/r4t_trace_$f -type f -size +4M -delete
done 2>&1 | tee $OUT/r4t_gn_bwd_kernels.txt
