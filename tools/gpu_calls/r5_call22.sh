#!/bin/bash
# round 5, call 22: kernels per training step (rocprofv3 --kernel-trace of bench.py --train-only; the step is one hipGraph replay)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof_r5train
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof_r5train -o train --output-format csv -- python $ROOT/bench.py --train-only --train-steps 4 --train-warmup 2 --no-roofline > $OUT/prof_r5train.log 2>&1
echo "rc=$?"
python $ROOT/tools/train_step_kernel_count.py $OUT/prof_r5train | tee $OUT/r5u_train_step_kernels.txt
find $OUT/prof_r5train -type f -size +8M -delete
