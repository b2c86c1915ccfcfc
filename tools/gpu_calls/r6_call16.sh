#!/bin/bash
# round 6, call 16: attention forward alone, both kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6p_attn_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_attn -o attn -- python $ROOT/tools/attn_bench.py > /dev/null 2>&1
find $OUT/prof_attn -name "*kernel_stats*" | head -1 | xargs head -8 | cut -c1-220 | tee -a $OUT/r6p_attn_bench.txt
