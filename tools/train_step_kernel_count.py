#!/usr/bin/env python
"""Kernels per DSM training step from a rocprofv3 --kernel-trace of `bench.py --train-only`: dispatches between consecutive
adam_kernel launches (one per step), their busy time and span.  usage: train_step_kernel_count.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import Counter

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "adam_kernel" in n]
print("%d dispatches, %d optimizer steps" % (len(rows), len(idx)))
for a, b in list(zip(idx[:-1], idx[1:]))[-4:]:
    seg = rows[a + 1:b + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    small = [r for r in seg if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 13000]
    print("step: %d kernels, busy %.2f ms, span %.2f ms; %d kernels under 13 us = %.2f ms"
          % (len(seg), busy, span, len(small), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small) / 1e6))
seg = rows[idx[-2] + 1:idx[-1] + 1]
short = lambda n: n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48]   # noqa: E731
c, t = Counter(), Counter()
for r in seg:
    c[short(r["Kernel_Name"])] += 1
    t[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in t.most_common(30):
    print("  %-48s %4d launches %8.3f ms" % (k, c[k], v / 1e6))
