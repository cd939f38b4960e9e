#!/usr/bin/env python3
"""R&D: A/B of execution modes inside ONE process (same box, same
allocations, interleaved rounds): kernel ms per launch by HIP events.

    python tools/ab_inproc.py [--nvel 19] [--size 256 256 256] [--hydro 1]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402
import torch                # noqa: E402

import ludwig_amd           # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvel", type=int, default=19)
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--hydro", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--modes", default="fused,fused_soa")
    args = ap.parse_args()
    names = args.modes.split(",")
    modes = {"fused": ludwig_amd.FUSED, "fused_soa": ludwig_amd.FUSED_SOA,
             "inplace": ludwig_amd.INPLACE}
    lbs = []
    for nm in names:
        # "fused+key=value+..." applies lbmi_tune settings to that handle
        base, *tunes = nm.split("+")
        lb = ludwig_amd.LB(args.nvel, tuple(args.size), 1, mode=modes[base])
        for kv in tunes:
            k, v = kv.split("=")
            lb.tune(k, int(v))
        lb.relaxation_set("m10", 0.1, 0.3)
        m = ludwig_amd.model(args.nvel)
        synthetic.fill_device(lb, m["cv"], m["wv"], tuple(args.size))
        hy = ludwig_amd.Hydro(lb.nall, lb.device,
                              force=np.zeros((3,) + lb.nall)) if args.hydro else None
        for _ in range(5):
            lb.step(hy)
        lb.synchronize()
        lbs.append((nm, lb, hy))
    sites = args.size[0] * args.size[1] * args.size[2]
    bpl = 2 * 8 * args.nvel + (56 if args.hydro else 0)
    res = {nm: [] for nm in names}
    for r in range(args.rounds):
        for nm, lb, hy in lbs:
            lb.timing(True)
            for _ in range(args.steps):
                lb.step(hy)
            lb.synchronize()
            kms, n = lb.timing_read()
            lb.timing(False)
            res[nm].append(kms / n)
    for nm in names:
        v = np.array(res[nm])
        print("%-10s hydro=%d  ms/launch: %s  median %.4f  -> %.0f GB/s (%d B/LUP)"
              % (nm, args.hydro, " ".join("%.4f" % x for x in v), np.median(v),
                 1e-6 * bpl * sites / np.median(v), bpl))


if __name__ == "__main__":
    main()
