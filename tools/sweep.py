#!/usr/bin/env python3
"""R&D: sweep the run-time launch tuning of the fused kernel on one box.

usage: tools/sweep.py [--hydro 0|1] [--steps N] g=0,8,16 lds=0,40960,65536
Prints kernel ms (HIP events) for every combination, in one process.
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import ludwig_amd  # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def main():
    hydro_on, steps, nvel = 0, 30, 19
    mode_name = "fused"
    size = (256, 256, 256)
    grid = {"g": [0], "lds": [0]}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--hydro":
            hydro_on = int(args.pop(0))
        elif a == "--steps":
            steps = int(args.pop(0))
        elif a == "--mode":
            mode_name = args.pop(0)
        elif a == "--nvel":
            nvel = int(args.pop(0))
        elif a == "--size":
            size = tuple(int(args.pop(0)) for _ in range(3))
        else:
            k, v = a.split("=")
            grid[k] = [int(x) for x in v.split(",")]
    mode = {"fused": ludwig_amd.FUSED, "inplace": ludwig_amd.INPLACE,
            "eager": ludwig_amd.EAGER}[mode_name]
    lb = ludwig_amd.LB(nvel, size, 1, mode=mode,
                       halo_scheme=ludwig_amd.HALO_REDUCED)
    lb.relaxation_set("m10", 0.1, 0.3)
    m = ludwig_amd.lb.model(nvel)
    synthetic.fill_device(lb, m["cv"], m["wv"], size)
    hydro = None
    if hydro_on:
        hydro = ludwig_amd.Hydro(lb.nall, lb.device)
        hydro.force = torch.zeros((3,) + lb.nall, dtype=torch.float64,
                                  device=lb.device)
        torch.cuda.synchronize()
    for _ in range(5):
        lb.step(hydro)
    sites = size[0] * size[1] * size[2]
    tag = os.path.basename(os.environ.get("LBMI_LIB", "default"))
    for rep in range(2):
        for g, lds in itertools.product(grid["g"], grid["lds"]):
            lb.tune("xcd_group", g)
            lb.tune("lds_cap", lds)
            for _ in range(3):
                lb.step(hydro)
            lb.synchronize()
            lb.timing(True)
            for _ in range(steps):
                lb.step(hydro)
            ms, n = lb.timing_read()
            lb.timing(False)
            t = ms / n
            bpl = 16 * nvel + (56 if hydro_on else 0)
            print("%-22s %s hydro=%d g=%-4d lds=%-6d  %.4f ms  %8.1f MLUPS  %6.0f GB/s(%dB)"
                  % (tag, mode_name, hydro_on, g, lds, t, sites / t * 1e-3,
                     bpl * 1e-9 * sites / (t * 1e-3), bpl), flush=True)
    lb.free()


if __name__ == "__main__":
    main()
