#!/bin/bash
# round 6, call 25 (and again as call 28 on the very last sources): the final sources -- the whole GPU suite, smoke, the default bench line (the driver's command), the per-launch
# list of one evaluation, rocprofv3 kernel stats + PMC passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | flt | tail -6 | tee $OUT/r6p_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | flt | tail -2 | tee $OUT/r6p_smoke.txt
timeout 1200 python bench.py > $OUT/r6p_bench_default.json 2> $OUT/r6p_bench_default.err
tail -c 600 $OUT/r6p_bench_default.json; tail -5 $OUT/r6p_bench_default.err
SSDE_MATRIX=bf16x6 timeout 600 python tools/op_times.py 2>&1 | flt > $OUT/r6p_op_times.txt
head -3 $OUT/r6p_op_times.txt
timeout 1500 bash tools/profile_gpu.sh r6final 2>&1 | tail -15
