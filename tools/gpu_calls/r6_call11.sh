#!/bin/bash
# round 6, call 11: timing probe -- what the 95 GroupNorm finalize launches cost inside the graph (the statistics are left unset: timing only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
F=$OUT/r6k_gn_finalize_probe.txt
: > $F
line() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
c = d['roofline']['by_class']
print('images/s %.3f  ms/iter %.2f  train %.5f s/step  3x3 %.2f ms gn %.2f ms (%d launches) sclk %.0f MHz %.0f W' % (d['value'], d['ms_per_step'], d['train']['value'], c['conv3x3_fused']['ms'], c.get('groupnorm_stats',{}).get('ms',0), c.get('groupnorm_stats',{}).get('launches',0), d['telemetry']['legs']['sampler']['sclk_mhz']['mean'], d['telemetry']['legs']['sampler']['power_w']['mean']))"; }
for rep in 1 2; do
  for W in product nofinalize; do
    [ $W = product ] && unset SSDE_PROBE_SKIP_GN_FINALIZE || export SSDE_PROBE_SKIP_GN_FINALIZE=1
    echo "== $W, bench pass $rep" >> $F
    timeout 600 python bench.py --no-cpu-baseline --no-extras --no-other-matrix --no-exchange-probe 2>$OUT/r6k_err_$W.txt | line >> $F
  done
done
cat $F
