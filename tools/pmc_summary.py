#!/usr/bin/env python3
"""Summarise tools/profile_bench.sh output into profiles/: the kernel-stats
CSV and a JSON with the HBM traffic per launch of the fused kernel
(FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md, HBM
section; here it is also the calibration on a known byte count, since the
algorithmic reads are a lower bound).

    python tools/pmc_summary.py <tag> <profiles-name> [kernel-substring]
"""

import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avg(tag, sub, counter, kernel):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, sub),
                                   "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter csv for " + sub
    vals = {}
    rows = [r for r in csv.DictReader(open(files[0]))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    # several instantiations of the kernel may have run (start-up steps, the
    # hydro-every-step leg): the one with the most dispatches is the timed loop
    names = {}
    for r in rows:
        names.setdefault(r["Kernel_Name"], set()).add(r["Dispatch_Id"])
    main_name = max(names, key=lambda k: len(names[k]))
    for row in rows:
        if row["Kernel_Name"] == main_name:
            key = row["Dispatch_Id"]
            vals[key] = vals.get(key, 0.0) + float(row["Counter_Value"])
    v = sorted(vals.values())
    # drop the first launches (warm-up: cold caches and page faults)
    v = list(vals.values())[2:]
    return len(v), sum(v) / len(v)


def source_sha1():
    """As bench.py:kernel_source_sha1: the bench quotes this file's traffic
    only while the kernel source is the one it was measured with."""
    import hashlib
    h = hashlib.sha1()
    for f in ("lbmi_kernels.hip", "lbmi_kernels.h"):
        with open(os.path.join(ROOT, "ludwig_amd", "csrc", f), "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


def main():
    tag, name = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "k_propagate_collide"
    bench = json.loads(open(os.path.join(ROOT, "gpurun_out", tag + "_bench.json"))
                       .read().strip().splitlines()[-1])
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "**",
                                   "*kernel_stats.csv"), recursive=True)
    assert stats
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"))
    kavg = None
    kname = None
    for row in csv.DictReader(open(stats[0])):
        if kernel in row["Name"] and (kavg is None or int(row["Calls"]) > kavg[1]):
            kavg = (float(row["AverageNs"]) * 1e-6, int(row["Calls"]))
            kname = row["Name"]
    nf, fetch = counter_avg(tag, "fetch", "FETCH_SIZE", kernel)
    nw, write = counter_avg(tag, "write", "WRITE_SIZE", kernel)
    rl = bench["roofline"]
    # the stats pass prints its own bench line: HIP events and rocprofv3 of
    # ONE process (separate processes differ by a few per cent through the
    # physical placement of their arrays)
    same = None
    try:
        log = open(os.path.join(ROOT, "gpurun_out", tag + "_stats.log")).read()
        line = [x for x in log.splitlines() if x.startswith("{")][-1]
        same = json.loads(line)["roofline"]["avg_launch_ms"]
    except Exception:
        pass
    algo = rl["bytes_per_lup"] * rl["lups_per_launch"]
    traffic = 2.0 * fetch * 1024.0 + write * 1024.0
    out = {
        "FETCH_SIZE": {"launches": nf, "avg_counter_KiB": fetch},
        "WRITE_SIZE": {"launches": nw, "avg_counter_KiB": write},
        "summary": {
            "kernel": kernel, "kernel_instance": kname,
            "kernel_source_sha1": source_sha1(),
            "workload": bench["config"]["workload"],
            "mode": bench["config"]["mode"], "order": bench["config"].get("order"),
            "fetch_bytes_corrected_x2": 2.0 * fetch * 1024.0,
            "write_bytes": write * 1024.0,
            "traffic_bytes": traffic,
            "algorithmic_bytes": algo,
            "bytes_per_lup": rl["bytes_per_lup"],
            "traffic_over_algorithmic": traffic / algo,
            "rocprofv3_avg_kernel_ms": kavg[0] if kavg else None,
            "rocprofv3_calls": kavg[1] if kavg else None,
            "bench_hip_event_avg_kernel_ms": rl["avg_launch_ms"],
            "hip_event_avg_kernel_ms_in_the_rocprofv3_stats_process": same,
            "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), "
                    "same command and same box as the stats pass and the bench "
                    "line; FETCH_SIZE doubled (gfx950); stores include the "
                    "zero-filled y/z halo lanes",
        },
    }
    with open(os.path.join(ROOT, "profiles", name + "_pmc_hbm_traffic.json"), "w") as fp:
        json.dump(out, fp, indent=1)
    shutil.copy(os.path.join(ROOT, "gpurun_out", tag + "_bench.json"),
                os.path.join(ROOT, "profiles", name + "_bench.json"))
    print(json.dumps(out["summary"], indent=1))


if __name__ == "__main__":
    main()
