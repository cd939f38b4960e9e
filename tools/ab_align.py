#!/usr/bin/env python3
"""R&D: does the relative placement of f and fprime matter? ONE allocation,
fprime carved at different offsets behind f, nt_store toggled per offset.

    python tools/ab_align.py [--nvel 19] [--size 256 256 256]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402
import torch                # noqa: E402

import ludwig_amd           # noqa: E402
from ludwig_amd import lib as _l  # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvel", type=int, default=19)
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--pads", type=int, nargs="+",
                    default=[0, 4096, 65536, 1 << 20, (1 << 21) + 4096,
                             5 << 20, (8 << 20) + 65536, 32 << 20])
    args = ap.parse_args()
    lb = ludwig_amd.LB(args.nvel, tuple(args.size), 1, mode=ludwig_amd.FUSED)
    lb.relaxation_set("m10", 0.1, 0.3)
    m = ludwig_amd.model(args.nvel)
    synthetic.fill_device(lb, m["cv"], m["wv"], tuple(args.size))
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
    n = args.nvel * lb.nsite
    f_init = lb.f.clone()
    big = torch.zeros(2 * n + (max(args.pads) // 8) + 1024, dtype=torch.float64,
                      device=lb.device)
    sites = args.size[0] * args.size[1] * args.size[2]
    bpl = 2 * 8 * args.nvel + 56
    for pad in args.pads:
        a = big[0:n].view((args.nvel,) + lb.nall)
        off = n + pad // 8
        b = big[off:off + n].view((args.nvel,) + lb.nall)
        a.copy_(f_init)
        torch.cuda.synchronize()
        _l.check(lb._lib.lbmi_lb_bind(lb._h, ctypes.c_void_p(a.data_ptr()),
                                      ctypes.c_void_p(b.data_ptr())))
        lb._a, lb._b = a, b
        out = []
        for nt in (0, 1, 0, 1):
            lb.tune("nt_store", nt)
            for _ in range(4):
                lb.step(hy)
            lb.synchronize()
            lb.timing(1)
            lb.run(hy, args.steps)
            lb.synchronize()
            kms, k = lb.timing_read()
            lb.timing(0)
            out.append(kms / k)
        print("pad %9d B  (fprime - f) mod 2MiB = %8d  nt0 %.4f %.4f  nt1 %.4f %.4f ms  -> %.0f / %.0f GB/s"
              % (pad, (b.data_ptr() - a.data_ptr()) % (1 << 21), out[0], out[2], out[1], out[3],
                 1e-6 * bpl * sites / min(out[0], out[2]), 1e-6 * bpl * sites / min(out[1], out[3])))
        lb.lb_flush()
        lb.synchronize()


if __name__ == "__main__":
    main()
