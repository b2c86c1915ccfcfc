#!/bin/bash
# round 6, call 35: the N > 1 code paths on the final sources -- two ranks on one device over gloo (RCCL refuses two ranks per
# device): the sampler replicas and the training step with its bucketed exchange; also the driver's own launch form
# (torch.distributed.run, one rank) to check the line it prints
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python bench.py --gpus 2 --dist-backend gloo --share-device --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-other-matrix --train-steps 10 --train-warmup 3 > $OUT/r6x_bench_2ranks_gloo_share_device.json 2> $OUT/r6x_bench_2ranks.err; echo "2-rank rc=$?"
tail -c 1200 $OUT/r6x_bench_2ranks_gloo_share_device.json; echo; tail -4 $OUT/r6x_bench_2ranks.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-other-matrix > $OUT/r6x_bench_torchrun_1rank.json 2> $OUT/r6x_bench_torchrun.err; echo "torchrun rc=$?"
python -c "
import json
for f in ('$OUT/r6x_bench_2ranks_gloo_share_device.json', '$OUT/r6x_bench_torchrun_1rank.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d.get('dist_backend'), d['train']['value'], d['train'].get('replicas_bit_identical_after_timed_steps'), d['train'].get('path'))"
