"""Kernels of one steady-state step from a rocprofv3 kernel trace: the launches
between two successive launches of the kernel named (default: the 40th and
41st), with start, end, duration in us.
Usage: python tools/step_from_trace.py <dir> [kernel-substring [which]]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "k_propagate_collide"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
    a, b = idx[which], idx[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%9.1f %9.1f %8.1f  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    print("# start to start: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main()
