#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests/ -m gpu -x -q > $OUT/r3t_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/r3t_pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
