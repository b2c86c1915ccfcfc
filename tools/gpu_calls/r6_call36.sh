#!/bin/bash
# round 6, call 36: the driver's launch form with ONE rank (torch.distributed.run --nproc-per-node 1): the one-rank exchange probe
# must not wait for the agent's store
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-other-matrix > $OUT/r6y_bench_torchrun_1rank.json 2> $OUT/r6y_bench_torchrun.err; echo "torchrun rc=$?"
grep -v '^frame #' $OUT/r6y_bench_torchrun.err | tail -6 | cut -c1-300
python -c "
import json
d = json.loads(open('$OUT/r6y_bench_torchrun_1rank.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['train']['value'], d['train'].get('exchange_one_rank'))"
