#!/bin/bash
# round 5, call 19: the whole GPU suite (both matrix modes) + smoke on the final sources; two ranks on one GPU over gloo (the N > 1
# code paths of the training step with the round's kernels: program runs + bucketed exchange instead of the step graph)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r5r_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/r5r_smoke.txt
timeout 600 python bench.py --gpus 2 --dist-backend gloo --share-device --train-only --train-steps 10 --train-warmup 3 > $OUT/r5r_bench_2ranks_gloo_share_device.json 2> $OUT/r5r_bench_2ranks.err; echo "2-rank rc=$?"
tail -c 1200 $OUT/r5r_bench_2ranks_gloo_share_device.json; tail -3 $OUT/r5r_bench_2ranks.err
