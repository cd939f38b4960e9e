#!/usr/bin/env python3
"""The last steps of a rocprofv3 kernel trace as a timeline (start, end,
duration in microseconds, queue, kernel): tools/timeline.py <trace.csv> [nrows [skip]]
(skip: rows to leave out at the end, e.g. the flush and the moments of a bench run)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[-(n + skip):len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
print("#   start us    end us     dur us")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%10.1f %10.1f %9.1f   q=%s %s" % (s * 1e-3, e * 1e-3, (e - s) * 1e-3,
                                            r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
