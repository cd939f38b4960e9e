#!/usr/bin/env python3
"""Device time of the rows "next" that bench.py does not cover: the step with
flat walls (lb_collide, lb_halo, wall_bbl, lb_propagation) in EAGER and
FUSED_HALO mode, with and without slip, and the two-distribution step
(phi_lb_to_field, field_halo, field_grad, lb_collide(binary), lb_halo,
lb_propagation; EAGER). Wall clock around synchronised loops, ms/step.

    python tools/bench_next_rows.py [--size 256 256 256] [--steps 40]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402
import torch                # noqa: E402

import ludwig_amd           # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def timed(fn, lb, steps):
    for _ in range(5):
        fn()
    lb.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    lb.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def walls(args, mode, slip):
    n = tuple(args.size)
    lb = ludwig_amd.LB(19, n, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    m = ludwig_amd.model(19)
    synthetic.fill_device(lb, m["cv"], m["wv"], n)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall),
                          status=np.zeros(lb.nall, dtype=np.int8))
    torch.cuda.synchronize()
    lb.wall_map((0, 0, 1), hy.status)
    nlink = lb.wall_links_build(hy.status, (0, 0, 1))
    if slip:
        lb.wall_slip_set(hy.status, (0, 0, 0.5), (0, 0, 0.5))
    else:
        lb.wall_velocity_set((0.01, 0, 0), (-0.01, 0, 0))

    def step():
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        lb.lb_propagation()

    ms = timed(step, lb, args.steps)
    t_bbl = timed(lb.wall_bbl, lb, args.steps) if mode == ludwig_amd.EAGER else None
    lb.free()
    sites = n[0] * n[1] * n[2]
    print("walls z, %s, %s: %d links, %.4f ms/step = %.0f MLUPS%s"
          % ({0: "EAGER", 3: "FUSED_HALO"}[mode], "slip 0.5" if slip else "moving, no slip",
             nlink, ms, 1e-3 * sites / ms,
             "" if t_bbl is None else " (wall_bbl alone %.4f ms)" % t_bbl), flush=True)


def binary(args, mode):
    n = tuple(args.size)
    lb = ludwig_amd.LB(19, n, 2, ndist=2, mode=mode)  # nhalo 2: field gradients
    lb.relaxation_set("m10", 0.1, 0.3)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
    dev = lb.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    m = ludwig_amd.model(19)
    w = torch.tensor(m["wv"], dtype=torch.float64, device=dev)
    f = lb._a.view((2, 19) + lb.nall)                # the handle's own array
    phi0 = 0.05 * (torch.rand(lb.nall, dtype=torch.float64, device=dev, generator=g) - 0.5)
    f[0] = w.reshape(19, 1, 1, 1)
    f[1, 0] = phi0                                   # all of phi in g_0 (phi_lb_from_field)
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=dev)
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=dev)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        lb.phi_to_field(phi)
        lb.field_halo_n(phi, 2)
        lb.field_grad_7pt(phi, grad, delsq)
        lb.lb_collide_binary(hy, -0.00625, 0.00625, 0.004, 1.25, phi, grad, delsq)
        lb.lb_halo()
        lb.lb_propagation()

    ms = timed(step, lb, args.steps)
    sites = n[0] * n[1] * n[2]
    if mode != ludwig_amd.EAGER:
        print("two distributions (symmetric_lb), %s: %.4f ms/step = %.0f MLUPS"
              % ({3: "FUSED_HALO", 1: "FUSED"}[mode], ms, 1e-3 * sites / ms), flush=True)
        lb.free()
        return
    parts = {}
    for name, fn in (("phi_lb_to_field", lambda: lb.phi_to_field(phi)),
                     ("field_halo", lambda: lb.field_halo_n(phi, 2)),
                     ("field_grad", lambda: lb.field_grad_7pt(phi, grad, delsq)),
                     ("lb_collide(binary)", lambda: lb.lb_collide_binary(
                         hy, -0.00625, 0.00625, 0.004, 1.25, phi, grad, delsq)),
                     ("lb_halo", lb.lb_halo),
                     ("lb_propagation", lb.lb_propagation)):
        parts[name] = timed(fn, lb, args.steps)
    sites = n[0] * n[1] * n[2]
    print("two distributions (symmetric_lb), EAGER: %.4f ms/step = %.0f MLUPS; %s"
          % (ms, 1e-3 * sites / ms,
             ", ".join("%s %.3f" % kv for kv in parts.items())), flush=True)
    lb.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    for mode in (ludwig_amd.EAGER, ludwig_amd.FUSED_HALO):
        for slip in (0, 1):
            walls(args, mode, slip)
    binary(args, ludwig_amd.EAGER)
    binary(args, ludwig_amd.FUSED_HALO)
    binary(args, ludwig_amd.FUSED)


if __name__ == "__main__":
    main()
