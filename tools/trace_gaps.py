#!/usr/bin/env python3
"""Kernel durations and start-to-start intervals from a rocprofv3 kernel trace
(*_kernel_trace.csv): tools/trace_gaps.py <csv> [name-substring]"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "k_propagate_collide"
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows
     if sub in r["Kernel_Name"]]
k.sort()
s = np.array([a for a, b in k], dtype=np.int64)
e = np.array([b for a, b in k], dtype=np.int64)
dur = (e - s) * 1e-3
gap = (s[1:] - e[:-1]) * 1e-3
n = len(k)
tail = slice(n // 2, None)            # the timed region is the second half
print("%s: %d launches; duration us: median %.2f  mean %.2f  min %.2f" %
      (sub, n, np.median(dur[tail]), dur[tail].mean(), dur[tail].min()))
print("gap end->next start us: median %.2f  mean %.2f  p90 %.2f" %
      (np.median(gap[tail]), gap[tail].mean(), np.percentile(gap[tail], 90)))
print("start->start us: median %.2f" % np.median(np.diff(s)[tail] * 1e-3))
