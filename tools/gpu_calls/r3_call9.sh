#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" > $OUT/r3i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3i_pytest.log
for M in 4 2; do echo "== SSDE_WGRAD_WINOGRAD=$M"; SSDE_WGRAD_WINOGRAD=$M timeout 300 python tools/wgrad_bench.py 128 2>&1 | grep -v amdgpu | grep "pro=2"; done | tee $OUT/r3i_wgrad_bench.txt
cd /tmp && export TMPDIR=/tmp
SSDE_WGRAD_WINOGRAD=4 rocprofv3 --kernel-trace --stats -d $OUT/prof_wg4 -o wg --output-format csv -- python $ROOT/tools/wgrad_bench.py 128 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/prof_wg4/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-70s calls %5s avg %9.1f us %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
