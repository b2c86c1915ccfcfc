#!/bin/bash
# round 6, call 21: is test_groupnorm_statistics_merged_by_the_transform_pass[f32] flaky?  six runs of it, then the rest of the suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
: > $OUT/r6l_gn_merge_repeat.txt
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_train_gpu.py -q -k "merged_by_the_transform_pass" 2>&1 | flt | grep -E "^E  |passed|failed" | tr '\n' ' ' >> $OUT/r6l_gn_merge_repeat.txt
  echo >> $OUT/r6l_gn_merge_repeat.txt
done
cat $OUT/r6l_gn_merge_repeat.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | flt | tail -15 | tee $OUT/r6l_gpu_suite.txt
