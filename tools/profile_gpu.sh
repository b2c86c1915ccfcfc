#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of bench.py and separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy) as MI355X_MICROARCH.md prescribes (one counter group per pass, --kernel-trace only).
# Usage: tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/...  (summaries are copied to profiles/ afterwards)
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --prewarm-s 0 --no-telemetry --no-exchange-probe --no-other-matrix --no-cpu-baseline --no-roofline --no-extras --train-steps 3 --train-warmup 1"
echo "== kernel trace (sampler + train step)"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $BENCH > $OUT/trace.log 2>&1
echo "rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o bench --output-format csv -- $BENCH --no-train > $OUT/pmc_$C.log 2>&1
  echo "rc=$?"
done
echo "== pmc MFMA busy"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_MFMA -o bench --output-format csv -- $BENCH --no-train > $OUT/pmc_MFMA.log 2>&1
echo "rc=$?"
find $OUT -name "*.csv" | head -40
python $ROOT/tools/summarize_profile.py $OUT $ROOT/gpurun_out/profile_summary_$TAG.json
# keep the merge small: drop the raw per-dispatch files above 8 MB
find $OUT -type f -size +8M -delete
