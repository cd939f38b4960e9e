"""The single-fluid collision with the relaxed symmetric stress (fe->use_stress_relaxation, lbmi_lb_collide_fe) at
256^3, ms per step (lb_collide_fe + lb_halo + lb_propagation) per execution mode.
Usage: python tools/bench_relax.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch                              # noqa: E402

import ludwig_amd                         # noqa: E402
from ludwig_amd import synthetic          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    m = ludwig_amd.lb.model(19)
    for name, mode in (("eager", ludwig_amd.EAGER), ("fused_halo", ludwig_amd.FUSED_HALO),
                       ("fused", ludwig_amd.FUSED)):
        lb = ludwig_amd.LB(19, (n, n, n), 1, mode=mode)
        lb.relaxation_set("m10", 0.1, 0.3)
        synthetic.fill_device(lb, m["cv"], m["wv"], (n, n, n))
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        phi = 0.1 * torch.randn(lb.nall, dtype=torch.float64, device=lb.device)
        grad = 0.01 * torch.randn((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = 0.01 * torch.randn(lb.nall, dtype=torch.float64, device=lb.device)
        torch.cuda.synchronize()

        def step():
            lb.lb_collide_fe(hy, -0.00625, 0.00625, 0.004, phi, grad, delsq)
            lb.lb_halo()
            lb.lb_propagation()

        for _ in range(5):
            step()
        lb.synchronize()
        t0 = time.perf_counter()
        k = 40
        for _ in range(k):
            step()
        lb.synchronize()
        dt = (time.perf_counter() - t0) / k
        print("%-10s %.4f ms per step  %.0f MLUPS" % (name, 1e3 * dt, 1e-6 * n ** 3 / dt), flush=True)
        lb.free()


if __name__ == "__main__":
    main()
