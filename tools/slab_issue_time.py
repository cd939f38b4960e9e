"""Is the slab step bound by the host that issues it? One 32x256x256 slab on a
1-rank RCCL ring (as bench.py --selfring 1): the time lbmi_lb_run takes to
RETURN (everything issued) beside the time until the device has finished.
Usage: python tools/slab_issue_time.py [steps] [cartdim]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch                              # noqa: E402

import ludwig_amd                         # noqa: E402
from ludwig_amd import synthetic          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n = [256, 256, 256]
    n[dim] = 32
    for selfring in (0, 1):
        lb = ludwig_amd.LB(19, tuple(n), 1, mode=ludwig_amd.FUSED, cartdim=dim,
                           halo_scheme=ludwig_amd.HALO_REDUCED)
        lb.relaxation_set("m10", 0.1, 0.3)
        if selfring:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
        m = ludwig_amd.lb.model(19)
        synthetic.fill_device(lb, m["cv"], m["wv"], tuple(n), xrange=(0, n[0]))
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        hy.force = torch.empty((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        torch.cuda.synchronize()
        lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))
        lb.tune("hydro_lazy", 1)
        lb.run(hy, 20)
        lb.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            lb.run(hy, steps)
            t1 = time.perf_counter()
            lb.synchronize()
            t2 = time.perf_counter()
            print("selfring %d cartdim %d: issued in %.4f ms/step, finished in %.4f ms/step"
                  % (selfring, dim, 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps), flush=True)
        lb.free()


if __name__ == "__main__":
    main()
