#!/usr/bin/env python3
"""Average per dispatch of every counter of tools/profile_pmc.sh for the kernel
instance (name containing the substring) with the most dispatches:
    python tools/pmc_table.py <tag> <kernel-substring>"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sub = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_pmc*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if sub in r["Kernel_Name"]]
    names = {}
    for r in rows:
        names.setdefault(r["Kernel_Name"], set()).add(r["Dispatch_Id"])
    if not names:
        continue
    main = max(names, key=lambda k: len(names[k]))
    acc = {}
    for r in rows:
        if r["Kernel_Name"] != main:
            continue
        acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print("# %s  (%d dispatches)" % (main[:100], len(names[main])))
    for c, v in acc.items():
        vals = list(v.values())[1:] or list(v.values())
        print("%-32s %18.1f" % (c, sum(vals) / len(vals)))
