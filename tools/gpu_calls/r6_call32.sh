#!/bin/bash
# round 6, call 32: ssde_gn_finalize with a workgroup per group where a group has more than 256 entries (the 128x128 / 256x256
# levels of FFHQ-256: 109 launches had gone from 1.4 to 3.2 ms per evaluation with the 16-lane teams): parity, FFHQ-256 line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py -q -k "groupnorm or ffhq or finalize" 2>&1 | flt | tail -4 | tee $OUT/r6v_gn_big_groups_parity.txt
timeout 600 python bench.py --workload ffhq256 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r6v_bench_ffhq256.json 2> $OUT/r6v_bench_ffhq256.err; echo "ffhq rc=$?"
python -c "
import json
d = json.loads(open('$OUT/r6v_bench_ffhq256.json').read().strip().splitlines()[-1])
print('ffhq256 images/s %.5f  ms/iter %.2f' % (d['value'], d['ms_per_step']))
for k, v in d['roofline']['by_class'].items(): print(k, v.get('launches'), round(v['ms'], 3), v.get('frac'))" | tee $OUT/r6v_bench_ffhq256.txt
