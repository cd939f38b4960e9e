#!/usr/bin/env python3
"""R&D: device time of the auxiliary kernels at 256^3 (record stream pack /
unpack, hydro_field_set, moments, halo schemes, eager stages)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import ludwig_amd  # noqa: E402
from ludwig_amd import lib as L  # noqa: E402
from ludwig_amd import synthetic  # noqa: E402


def timed(lb, fn, reps=20):
    for _ in range(3):
        fn()
    lb.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    lb.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    nvel, size = 19, (256, 256, 256)
    lb = ludwig_amd.LB(nvel, size, 1)
    m = ludwig_amd.lb.model(nvel)
    synthetic.fill_device(lb, m["cv"], m["wv"], size)
    sites = size[0] * size[1] * size[2]
    rec = torch.empty(size + (nvel,), dtype=torch.float64, device=lb.device)
    u = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    torch.cuda.synchronize()
    import ctypes
    p = ctypes.c_void_p

    def gbs(nbytes, ms):
        return nbytes / ms * 1e-6

    t = timed(lb, lambda: L.check(lb._lib.lbmi_lb_records_pack(lb._h, p(rec.data_ptr()))))
    print("records_pack      %.3f ms  %6.0f GB/s" % (t, gbs(2 * 152 * sites, t)))
    t = timed(lb, lambda: L.check(lb._lib.lbmi_lb_records_unpack(lb._h, p(rec.data_ptr()))))
    print("records_unpack    %.3f ms  %6.0f GB/s" % (t, gbs(2 * 152 * sites, t)))
    t = timed(lb, lambda: lb.hydro_field_set(u, (0, 0, 0)))
    print("hydro_field_set   %.3f ms  %6.0f GB/s" % (t, gbs(24 * lb.nsite, t)))
    t = timed(lb, lambda: lb.moments_of(lb.f), reps=5)
    print("moments           %.3f ms  %6.0f GB/s (incl. D2H of 9 doubles)" % (t, gbs(152 * sites, t)))
    t = timed(lb, lambda: lb.halo(lb.f, 0))
    print("halo FULL         %.3f ms" % t)
    t = timed(lb, lambda: lb.halo(lb.f, 2))
    print("halo REDUCED      %.3f ms" % t)
    t = timed(lb, lambda: lb.collide(lb.f, hy))
    print("k_collide (hydro) %.3f ms  %6.0f GB/s" % (t, gbs(360 * sites, t)))
    t = timed(lb, lambda: lb.propagate(lb.f, lb.fprime))
    print("k_propagate       %.3f ms  %6.0f GB/s" % (t, gbs(304 * sites, t)))
    lb.free()

    # row f2: symmetric free-energy force chain (nhalo = 2)
    for n in (128, 256):
        lb = ludwig_amd.LB(19, (n, n, n), 2)
        phi = 0.3 * torch.randn(lb.nall, dtype=torch.float64, device=lb.device)
        grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        force = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        torch.cuda.synchronize()
        s3 = n ** 3
        t = timed(lb, lambda: lb.field_halo_n(phi, 2))
        print("%d^3 field_halo(2)        %.4f ms" % (n, t))
        t = timed(lb, lambda: lb.field_grad_7pt(phi, grad, delsq))
        print("%d^3 grad_7pt             %.4f ms  %6.0f GB/s (40 B/site)" % (n, t, gbs(40 * s3, t)))
        t = timed(lb, lambda: lb.symmetric_force(-0.00625, 0.00625, 0.004, phi, force, grad, delsq))
        print("%d^3 force from grad      %.4f ms  %6.0f GB/s (88 B/site)" % (n, t, gbs(88 * s3, t)))
        t = timed(lb, lambda: lb.symmetric_force(-0.00625, 0.00625, 0.004, phi, force))
        print("%d^3 force from phi       %.4f ms  %6.0f GB/s (56 B/site)" % (n, t, gbs(56 * s3, t)))
        u = 0.01 * torch.randn((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        out = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        torch.cuda.synchronize()
        t = timed(lb, lambda: lb.cahn_hilliard(-0.00625, 0.00625, 0.004, 1.25, phi, u, out))
        print("%d^3 cahn_hilliard (phi)  %.4f ms  %6.0f GB/s (40 B/site)" % (n, t, gbs(40 * s3, t)))
        t = timed(lb, lambda: lb.symmetric_step(-0.00625, 0.00625, 0.004, 1.25, phi, u, force, out))
        print("%d^3 force + CH, one pass %.4f ms  %6.0f GB/s (88 B/site)" % (n, t, gbs(88 * s3, t)))
        lb.free()


if __name__ == "__main__":
    main()
