#!/usr/bin/env python3
"""R&D: does the relative placement of f and fprime in HBM matter?

Carves both arrays out of ONE allocation with a chosen gap between them and
times the fused kernel for each gap (same box, same process).
usage: tools/sweep_alloc.py pad_bytes...
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import ludwig_amd  # noqa: E402
from ludwig_amd import lib as L  # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def main():
    pads = [int(x) for x in sys.argv[1:]] or [0]
    nvel, size, steps = 19, (256, 256, 256), 40
    lb = ludwig_amd.LB(nvel, size, 1, mode=ludwig_amd.FUSED,
                       halo_scheme=ludwig_amd.HALO_REDUCED)
    lb.relaxation_set("m10", 0.1, 0.3)
    m = ludwig_amd.lb.model(nvel)
    nel = nvel * lb.nsite
    maxpad = max(pads)
    big = torch.zeros(2 * nel + maxpad // 8 + 64, dtype=torch.float64,
                      device=lb.device)
    torch.cuda.synchronize()
    sites = size[0] * size[1] * size[2]
    for rep in range(2):
        for pad in pads:
            a = big[:nel].view((nvel,) + lb.nall)
            o = nel + pad // 8
            b = big[o:o + nel].view((nvel,) + lb.nall)
            lb._a, lb._b = a, b
            L.check(lb._lib.lbmi_lb_bind(lb._h, ctypes.c_void_p(a.data_ptr()),
                                         ctypes.c_void_p(b.data_ptr())))
            synthetic.fill_device(lb, m["cv"], m["wv"], size)
            for _ in range(4):
                lb.step(None)
            lb.synchronize()
            lb.timing(True)
            for _ in range(steps):
                lb.step(None)
            ms, n = lb.timing_read()
            lb.timing(False)
            t = ms / n
            print("pad=%-10d  a%%2MiB=%-8d b-a mod 1MiB=%-8d  %.4f ms  %8.1f MLUPS"
                  % (pad, a.data_ptr() % (2 << 20),
                     (b.data_ptr() - a.data_ptr()) % (1 << 20), t,
                     sites / t * 1e-3), flush=True)
    lb.free()


if __name__ == "__main__":
    main()
