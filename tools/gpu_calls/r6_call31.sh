#!/bin/bash
# round 6, call 31: the FFHQ-256 bench line with its own roofline object; the double backward of op.fused_leaky_relu on the GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 600 python -m pytest tests/test_train_gpu.py -q -k "op_package or deferred" 2>&1 | flt | tail -3 | tee $OUT/r6u_op_package.txt
timeout 600 python bench.py --workload ffhq256 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r6u_bench_ffhq256.json 2> $OUT/r6u_bench_ffhq256.err; echo "ffhq rc=$?"
python -c "
import json
d = json.loads(open('$OUT/r6u_bench_ffhq256.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config'])
for k, v in d['roofline']['by_class'].items(): print(k, v.get('launches'), round(v['ms'], 3), v.get('frac'))
print({k: v for k, v in d['roofline'].items() if k not in ('by_class', 'note', 'kernel')})"
