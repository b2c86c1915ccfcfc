#!/usr/bin/env python
"""The CIFAR-10 PC sampler at batch 16 / 64 / 256 under the production kernel heuristic (engine.Lowering.wino_ok, thresholds
derived from the device's CU count): ms per PC iteration, images/s, and which 3x3 kernel the 93 launches of one evaluation
got -- against the same batch with every 3x3 layer forced to the direct kernel (SSDE_WINOGRAD=0) and to F(2x2,3x3) wherever
legal (SSDE_WINOGRAD=2).  reference: sampling.py:390-409 takes any `shape`.  GPU only; development tool."""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    batches = [int(v) for v in sys.argv[1:]] or [16, 64, 256]
    for b in batches:
        row = []
        for mode, label in (("1", "heuristic"), ("0", "direct"), ("2", "F(2x2) where legal"), ("4", "F(4x4) where legal")):
            env = dict(os.environ, SSDE_WINOGRAD=mode)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(b), "--steps", "10", "--warmup", "3",
                                "--no-cpu-baseline", "--no-extras", "--no-train"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                row.append("%s: failed (%s)" % (label, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"))
                continue
            d = json.loads(line[-1])
            row.append("%s: %.2f ms/iteration, %.3f images/s [%s]" % (label, d["ms_per_step"], d["value"], d["roofline"]["kernel"].split(": ", 1)[1]))
        print("batch %3d\n   " % b + "\n   ".join(row), flush=True)
