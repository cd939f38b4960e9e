#!/usr/bin/env python3
"""R&D: two ranks of an X-slab decomposition sharing ONE GPU, exchanging
over a real 2-rank RCCL communicator (if RCCL accepts two ranks on one
device), compared with the single-domain run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29621 tools/two_rank_one_gpu.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

import ludwig_amd                       # noqa: E402
from oracle import lb_oracle as lbo     # noqa: E402


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nvel, ntotal, nsteps = 19, (20, 14, 14), 6
    results = {}
    for mode in (ludwig_amd.EAGER, ludwig_amd.FUSED, ludwig_amd.FUSED_SOA):
        dec = ludwig_amd.SlabDecomposition(ntotal, world, rank)
        lb = ludwig_amd.LB(nvel, dec.nlocal, 1, mode=mode, halo_scheme=2,
                           device=0, cartsz=world, cartrank=rank)
        lb.relaxation_set("m10", 0.1, 0.3)
        ids = [ludwig_amd.LB.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        lb.comm_init(ids[0])
        p = lbo.make_param(nvel, dec.nlocal, 1, "m10", 0.1, 0.3)
        f0 = lbo.init_synthetic(p, ntotal, dec.noffset)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        for _ in range(nsteps):
            lb.step(hy)
        out = lb.lb_memcpy_d2h()[:, 1:-1, 1:-1, 1:-1].copy()
        parts = [None] * world
        dist.all_gather_object(parts, out)
        results[mode] = np.concatenate(parts, axis=1)
        lb.free()
        dist.barrier()
    if rank == 0:
        pg = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3)
        f = lbo.init_synthetic(pg)
        fp = np.zeros_like(f)
        for _ in range(nsteps):
            f, fp = lbo.step(pg, f, fp)
        ref = f[:, 1:-1, 1:-1, 1:-1]
        for mode, name in ((0, "eager"), (1, "fused"), (3, "fused_soa")):
            err = np.max(np.abs(results[mode] - ref)) / np.max(np.abs(ref))
            print("two ranks on one GPU, %s: max rel err vs single-domain oracle %.2e"
                  % (name, err), file=sys.stderr)
            assert err < 1e-12
        print("TWO-RANK RCCL EXCHANGE OK", file=sys.stderr)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
