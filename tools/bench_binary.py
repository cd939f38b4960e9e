#!/usr/bin/env python3
"""The two-distribution (free_energy symmetric_lb) step at 256^3 on one MI355X:
phi_lb_to_field, field_halo, field_grad_compute, lb_collide (binary), lb_halo,
lb_propagation through the C-ABI (the order of ludwig.c:558-860), in FUSED on
one GPU. --tune key=value,... (lbmi_tune: blocked, nt_store, xcd_group), so that
one process can be profiled per setting (tools/profile_binary.sh).

Algorithmic bytes per lattice update (D3Q19): the collision reads and writes
both distributions (2 x 304 B), reads phi, grad phi, delsq phi (40 B) and the
force (24 B), stores u (24 B) = 696 B; phi_lb_to_field reads the 19 g (152 B)
and stores phi (8 B) = 160 B; the gradients read phi (8 B) and store 32 B."""

import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ludwig_amd                                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--tune", default="")
    ap.add_argument("--mode", default="fused", choices=["fused", "fused_halo", "eager"])
    args = ap.parse_args()
    n = tuple(args.size)
    mode = {"fused": ludwig_amd.FUSED, "fused_halo": ludwig_amd.FUSED_HALO,
            "eager": ludwig_amd.EAGER}[args.mode]
    lb = ludwig_amd.LB(19, n, 2, ndist=2, mode=mode)         # nhalo 2: field gradients
    lb.relaxation_set("m10", 0.1, 0.3)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        lb.tune(k, int(v))
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
    dev = lb.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    m = ludwig_amd.model(19)
    w = torch.tensor(m["wv"], dtype=torch.float64, device=dev)
    f = lb._a.view((2, 19) + lb.nall)
    phi0 = 0.05 * (torch.rand(lb.nall, dtype=torch.float64, device=dev, generator=g) - 0.5)
    f[0] = w.reshape(19, 1, 1, 1)
    f[1, 0] = phi0
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

    for _ in range(5):
        step()
    lb.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    lb.synchronize()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    sites = n[0] * n[1] * n[2]
    print("two distributions (symmetric_lb) %dx%dx%d, %s, tune '%s', order of the deferred state %d: "
          "%.4f ms/step = %.0f MLUPS" % (*n, args.mode, args.tune, lb.state()[2], ms, 1e-3 * sites / ms),
          flush=True)
    lb.free()


if __name__ == "__main__":
    main()
