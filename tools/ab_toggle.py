#!/usr/bin/env python3
"""R&D: A/B of lbmi_tune settings on ONE handle (same arrays, same physical
placement), interleaved rounds: kernel ms per launch by HIP events.

    python tools/ab_toggle.py --key nt_store --values 0 1 [--hydro 1] [--nvel 19]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402

import ludwig_amd           # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvel", type=int, default=19)
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--hydro", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--key", default="nt_store")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--tune", default="")
    args = ap.parse_args()
    lb = ludwig_amd.LB(args.nvel, tuple(args.size), 1, mode=ludwig_amd.FUSED)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        lb.tune(k, int(v))
    lb.relaxation_set("m10", 0.1, 0.3)
    m = ludwig_amd.model(args.nvel)
    synthetic.fill_device(lb, m["cv"], m["wv"], tuple(args.size))
    hy = ludwig_amd.Hydro(lb.nall, lb.device,
                          force=np.zeros((3,) + lb.nall)) if args.hydro else None
    sites = args.size[0] * args.size[1] * args.size[2]
    bpl = 2 * 8 * args.nvel + (56 if args.hydro else 0)
    res = {v: [] for v in args.values}
    for r in range(args.rounds):
        for v in args.values:
            lb.tune(args.key, v)
            for _ in range(4):               # settle (order conversion etc.)
                lb.step(hy)
            lb.synchronize()
            lb.timing(1)
            for _ in range(args.steps):
                lb.step(hy)
            lb.synchronize()
            kms, n = lb.timing_read()
            lb.timing(0)
            res[v].append(kms / n)
    for v in args.values:
        a = np.array(res[v])
        print("%s=%d nvel=%d hydro=%d  ms/launch: %s  median %.4f -> %.0f GB/s (%d B/LUP)"
              % (args.key, v, args.nvel, args.hydro, " ".join("%.4f" % x for x in a),
                 np.median(a), 1e-6 * bpl * sites / np.median(a), bpl))


if __name__ == "__main__":
    main()
