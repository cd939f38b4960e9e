"""The N > 1 path on CPU: two gloo ranks, each owning an X slab.

What this covers (host logic shared with the product): the slab
decomposition (ludwig_amd.SlabDecomposition), the neighbour ring, which
plane goes to which neighbour's halo, the message lengths, and that the
slab-decomposed time step (X by exchange, Y/Z locally, in that order)
equals the single-domain step. The arithmetic is the oracle's; the device
kernels for the same exchange are tested in tests/test_gpu_parity.py.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ludwig_amd
from oracle import lb_oracle as lbo
from tests.common import interior


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _exchange_x(dec, f, nh):
    """Periodic ring: last interior plane -> next's low halo; first interior
    plane -> prev's high halo (halo_swap.c:762-784, 862-865). Full planes
    including the y/z halo extents."""
    lo_send = np.ascontiguousarray(f[:, nh])                  # first plane
    hi_send = np.ascontiguousarray(f[:, nh + dec.nlocal[0] - 1])
    lo_recv = np.empty_like(lo_send)
    hi_recv = np.empty_like(hi_send)
    assert lo_send[0].size == dec.plane_doubles(1)
    ops = [
        dist.P2POp(dist.isend, torch.from_numpy(hi_send), dec.next, tag=1),
        dist.P2POp(dist.irecv, torch.from_numpy(lo_recv), dec.prev, tag=1),
        dist.P2POp(dist.isend, torch.from_numpy(lo_send), dec.prev, tag=2),
        dist.P2POp(dist.irecv, torch.from_numpy(hi_recv), dec.next, tag=2),
    ]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    f[:, nh - 1] = lo_recv
    f[:, nh + dec.nlocal[0]] = hi_recv


def _worker(rank, world, port, nvel, ntotal, nsteps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nh = 1
        dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, nh)
        p = lbo.make_param(nvel, dec.nlocal, nh, "m10", 0.1, 0.3, 1.0,
                           (1e-6, 0.0, -1e-6))
        f = lbo.init_synthetic(p, ntotal, dec.noffset)
        fp = np.zeros_like(f)
        for _ in range(nsteps):
            lbo.collide(p, f)
            _exchange_x(dec, f, nh)      # X first ...
            lbo.halo_yz(p, f)            # ... then Y, Z over the full extent
            lbo.propagate(p, f, fp)
            f, fp = fp, f
        mo = torch.from_numpy(lbo.moments(p, f)[[0, 1, 5, 6, 7]].copy())
        dist.all_reduce(mo)
        q.put((rank, interior(f, nh).copy(), mo.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nvel", [19, 27])
def test_two_slabs_equal_single_domain(nvel):
    world, ntotal, nsteps = 2, (8, 5, 6), 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker,
                         args=(r, world, port, nvel, ntotal, nsteps, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, fi, mo = q.get(timeout=120)
        res[rank] = (fi, mo)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0

    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, (1e-6, 0.0, -1e-6))
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp)
    ref = interior(f, 1)
    got = np.concatenate([res[0][0], res[1][0]], axis=1)
    assert np.array_equal(got, ref)       # same arithmetic, same order
    mref = lbo.moments(p, f)[[0, 1, 5, 6, 7]]
    assert abs(res[0][1][1] - mref[1]) / mref[1] < 1e-14
    assert np.array_equal(res[0][1], res[1][1])
