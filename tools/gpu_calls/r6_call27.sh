#!/bin/bash
# round 6, call 27: ssde_colsum_finish with four samples of a lane in flight (same summation order): parity of the training
# chain, then the training step before / after (the round's previous library as the variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
flt() { grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | flt | tail -4 | tee $OUT/r6r_colsum_finish_parity.txt
F=$OUT/r6r_colsum_finish_ab.txt
: > $F
for rep in 1 2; do
  for V in before after; do
    [ $V = after ] && unset SSDE_LIB_PATH || export SSDE_LIB_PATH=$ROOT/tools/variants/libssde_hip_before_colsum_finish.so
    echo "== $V, pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6r_err_$V.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
t = d['train'] if 'train' in d else d
c = t.get('by_class', {})
print('train %.5f s/step  backward_elementwise %.2f ms (%d launches)' % (t['value'], c.get('backward_elementwise', {}).get('ms', -1), c.get('backward_elementwise', {}).get('launches', -1)))" >> $F 2>&1
  done
done
unset SSDE_LIB_PATH
cat $F
