"""The N > 1 path on CPU: gloo ranks, each owning an X slab, exchange their
boundary planes by executing THE PRODUCT'S schedule -- lbmi_x_schedule
(include/lbmi.h), the very list of sends and receives (peer, buffer, offset,
count, order) that liblbmi issues as ncclSend / ncclRecv inside one group
(lbmi_host.c: lbmi_x_ops -> lbmi_x_sendrecv). gloo's point-to-point matching
is RCCL's: the k-th send towards a peer meets the k-th receive from it; no
tags are used, so a schedule whose order does not pair up fails here -- also
in the two-rank case, where the previous and the next rank are the same peer
(test_a_wrong_order_is_noticed shows that it would).

What is the test's own: packing the planes into the staging buffers by their
documented layout ([k][plane site], components of the reduced or full
selection in p order) and the oracle's arithmetic for the rest of the step.
The device kernels behind the same schedule run in tests/test_gpu_ring.py.
Replaces the reference's halo_swap.c:762-881.
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

FBODY = (1e-6, 0.0, -1e-6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _selection(cv, reduced):
    if reduced:
        return ludwig_amd.SlabDecomposition.reduced_populations(cv)
    allp = list(range(len(cv)))
    return allp, allp


def _exchange_x(dec, f, nh, cv, reduced, packed, tamper=None):
    """One exchange of f (nvel, nall) along the decomposed direction by the
    product's schedule (X slabs, or dec.dim = 1, 2: slabs along Y, Z)."""
    nvel = f.shape[0]
    dim = dec.dim
    scheme = ludwig_amd.HALO_REDUCED if reduced else ludwig_amd.HALO_FULL
    ops = ludwig_amd.x_schedule(nvel, dec.nlocal, nh, dec.cartsz, dec.cartrank,
                                scheme=scheme, packed=packed, cartdim=dim)
    if tamper is not None:
        ops = tamper(ops)
    if reduced:
        lo, hi = ludwig_amd.SlabDecomposition.reduced_populations(cv, axis=dim)
    else:
        lo, hi = _selection(cv, False)        # fill a LOW halo / a HIGH halo
    first, last = nh, nh + dec.nlocal[dim] - 1
    psz = dec.plane_doubles(1)
    whole = f
    # the decomposed direction in front: f[k, plane] is plane `plane` of
    # component k, its sites in the order of the two remaining coordinates
    # (a view: what is stored through it lands in the array)
    f = np.moveaxis(whole, 1 + dim, 1)
    flat = whole.reshape(-1)
    assert flat.base is not None or flat is f  # a view: receives land in f
    buf = {"data": flat}
    if packed:
        # include/lbmi.h: SENDHI = last interior plane, the components that
        # fill a LOW halo, for the next rank; SENDLO = first interior plane,
        # the components that fill a HIGH halo, for the previous rank
        buf["sendhi"] = np.ascontiguousarray(f[lo, last]).reshape(-1)
        buf["sendlo"] = np.ascontiguousarray(f[hi, first]).reshape(-1)
        buf["recvlo"] = np.full(len(lo) * psz, np.nan)
        buf["recvhi"] = np.full(len(hi) * psz, np.nan)
        assert len(ops) == 4
    else:
        assert len(ops) == 2 * (len(lo) + len(hi))
    p2p = []
    for op in ops:
        t = torch.from_numpy(buf[op["buffer"]])[op["offset"]:op["offset"] + op["count"]]
        assert t.numel() == op["count"]
        fn = dist.isend if op["kind"] == "send" else dist.irecv
        p2p.append(dist.P2POp(fn, t, op["peer"]))
    for r in dist.batch_isend_irecv(p2p):
        r.wait()
    if packed:
        f[lo, first - 1] = buf["recvlo"].reshape(len(lo), *f.shape[2:])
        f[hi, last + 1] = buf["recvhi"].reshape(len(hi), *f.shape[2:])


def _worker(rank, world, port, nvel, ntotal, nsteps, reduced, packed, tamper, q, dim=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nh = 1
        dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, nh, dim=dim)
        cv = lbo.model(nvel)["cv"]
        p = lbo.make_param(nvel, dec.nlocal, nh, "m10", 0.1, 0.3, 1.0, FBODY)
        f = lbo.init_synthetic(p, ntotal, dec.noffset)
        fp = np.zeros_like(f)
        tf = _TAMPER[tamper] if (tamper and rank == 0) else None
        for _ in range(nsteps):
            lbo.collide(p, f)
            # X, Y, Z in this order, each over the full extent of the others
            # (halo_swap.c:709-1063); the decomposed one goes over the ring
            for d in range(3):
                if d == dim:
                    _exchange_x(dec, f, nh, cv, reduced, packed, tf)
                else:
                    lbo.halo_dirs(p, f, 1 << d)
            lbo.propagate(p, f, fp)
            f, fp = fp, f
        mo = torch.from_numpy(lbo.moments(p, f)[[0, 1, 5, 6, 7]].copy())
        dist.all_reduce(mo)
        q.put((rank, interior(f, nh).copy(), mo.numpy()))
    finally:
        dist.destroy_process_group()


def _swap_sends(ops):
    """A schedule that sends downwards before upwards on ONE side only."""
    sends = [k for k, op in enumerate(ops) if op["kind"] == "send"]
    ops = list(ops)
    ops[sends[0]], ops[sends[1]] = ops[sends[1]], ops[sends[0]]
    return ops


_TAMPER = {"swap_sends": _swap_sends}


def _run(world, nvel, ntotal, nsteps, reduced, packed, tamper=None, dim=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker,
                         args=(r, world, port, nvel, ntotal, nsteps, reduced,
                               packed, tamper, q, dim))
             for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, fi, mo = q.get(timeout=180)
        res[rank] = (fi, mo)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return res


def _single_domain(nvel, ntotal, nsteps):
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp)
    return p, f


@pytest.mark.parametrize("world,nvel,reduced,packed", [
    (2, 19, True, True),        # the FUSED step's exchange; prev == next
    (3, 19, True, True),
    (2, 27, True, True),
    (3, 27, False, True),       # lb_halo semantics: every population
    (2, 19, True, False),       # zero-copy: one message per component
    (3, 27, True, False),
    (2, 19, False, False),
])
def test_slabs_equal_single_domain(world, nvel, reduced, packed):
    ntotal, nsteps = ({2: 8, 3: 9}[world], 5, 6), 4
    res = _run(world, nvel, ntotal, nsteps, reduced, packed)
    p, f = _single_domain(nvel, ntotal, nsteps)
    got = np.concatenate([res[r][0] for r in range(world)], axis=1)
    assert np.array_equal(got, interior(f, 1))       # same arithmetic, same order
    mref = lbo.moments(p, f)[[0, 1, 5, 6, 7]]
    assert abs(res[0][1][1] - mref[1]) / mref[1] < 1e-14
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1])


@pytest.mark.parametrize("world,nvel,reduced,dim", [
    (2, 19, True, 2),           # grid 1_1_2: BASELINE config 3 as it is written (z slabs)
    (3, 19, False, 2),          # every population
    (2, 27, True, 2),
    (3, 19, True, 1),           # grid 1_3_1
    (2, 27, False, 1),
])
def test_slabs_along_y_or_z_equal_single_domain(world, nvel, reduced, dim):
    """Slabs along Y or Z (coords_rt.c:46-47, `grid 1_N_1`, `1_1_N`): the
    product's schedule with lbmi_options_t::cartdim -- packed messages of
    gathered planes (halo_swap.c:1074-1274), the local passes of the other two
    directions around the exchange in the order X, Y, Z -- reproduces the
    single domain bit for bit."""
    ntotal = [5, 6, 7]
    ntotal[dim] = {2: 8, 3: 9}[world]
    ntotal, nsteps = tuple(ntotal), 4
    res = _run(world, nvel, ntotal, nsteps, reduced, True, dim=dim)
    p, f = _single_domain(nvel, ntotal, nsteps)
    got = np.concatenate([res[r][0] for r in range(world)], axis=1 + dim)
    assert np.array_equal(got, interior(f, 1))
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1])


def _exchange_cart(dec, f, nh, cv, reduced, dim):
    """One exchange of f along direction `dim` of a Cartesian decomposition
    by the product's schedule (lbmi_x_schedule_dim): packed planes of the full
    extent of the other two directions, halos included."""
    nvel = f.shape[0]
    scheme = ludwig_amd.HALO_REDUCED if reduced else ludwig_amd.HALO_FULL
    ops = ludwig_amd.x_schedule(nvel, dec.nlocal, nh, dec.size, dec.rank, scheme=scheme,
                                packed=True, cartgrid=dec.grid, cartcoords=dec.coords,
                                dim=dim)
    if reduced:
        lo, hi = ludwig_amd.SlabDecomposition.reduced_populations(cv, axis=dim)
    else:
        lo = hi = list(range(nvel))
    first, last = nh, nh + dec.nlocal[dim] - 1
    g = np.moveaxis(f, 1 + dim, 1)
    psz = g.shape[2] * g.shape[3]
    buf = {"sendhi": np.ascontiguousarray(g[lo, last]).reshape(-1),
           "sendlo": np.ascontiguousarray(g[hi, first]).reshape(-1),
           "recvlo": np.full(len(lo) * psz, np.nan),
           "recvhi": np.full(len(hi) * psz, np.nan)}
    below, above = dec.neighbours(dim)
    assert len(ops) == 4 and {op["peer"] for op in ops} <= {below, above}
    p2p = []
    for op in ops:
        t = torch.from_numpy(buf[op["buffer"]])[op["offset"]:op["offset"] + op["count"]]
        assert t.numel() == op["count"]
        fn = dist.isend if op["kind"] == "send" else dist.irecv
        p2p.append(dist.P2POp(fn, t, op["peer"]))
    for r in dist.batch_isend_irecv(p2p):
        r.wait()
    g[lo, first - 1] = buf["recvlo"].reshape(len(lo), *g.shape[2:])
    g[hi, last + 1] = buf["recvhi"].reshape(len(hi), *g.shape[2:])


def _cart_worker(rank, grid, port, nvel, ntotal, nsteps, reduced, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    world = grid[0] * grid[1] * grid[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nh = 1
        dec = ludwig_amd.CartDecomposition(ntotal, grid, rank, nh)
        cv = lbo.model(nvel)["cv"]
        p = lbo.make_param(nvel, dec.nlocal, nh, "m10", 0.1, 0.3, 1.0, FBODY)
        f = lbo.init_synthetic(p, ntotal, dec.noffset)
        fp = np.zeros_like(f)
        for _ in range(nsteps):
            lbo.collide(p, f)
            # halo_swap.c:709-1063: X, Y, Z in this order, each over the full
            # extent of the others; a direction with one rank wraps locally
            for d in range(3):
                if grid[d] > 1:
                    _exchange_cart(dec, f, nh, cv, reduced, d)
                else:
                    lbo.halo_dirs(p, f, 1 << d)
            lbo.propagate(p, f, fp)
            f, fp = fp, f
        q.put((rank, dec.noffset, interior(f, nh).copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grid,nvel,reduced", [
    ((2, 2, 1), 19, False),     # lb_halo semantics: every population of every plane
    ((1, 2, 2), 19, True),      # the face-crossing populations of each pass are enough
    ((2, 1, 2), 27, True),
    ((2, 2, 2), 19, False),     # MPI_Dims_create's grid for eight ranks
])
def test_cartesian_decomposition_equals_single_domain(grid, nvel, reduced):
    """More than one direction decomposed (LBMI_CART_GENERAL; the reference's
    default for N ranks is MPI_Dims_create's grid): the product's schedule per
    decomposed direction, in the reference's order of passes, edges and
    corners completing through the halos of the earlier passes -- bit for bit
    the single domain."""
    ntotal, nsteps = (4 * grid[0], 3 * grid[1], 4 * grid[2]), 3
    world = grid[0] * grid[1] * grid[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cart_worker,
                         args=(r, grid, port, nvel, ntotal, nsteps, reduced, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    got = np.zeros((nvel,) + ntotal)
    for _ in range(world):
        rank, off, fi = q.get(timeout=240)
        got[:, off[0]:off[0] + fi.shape[1], off[1]:off[1] + fi.shape[2],
            off[2]:off[2] + fi.shape[3]] = fi
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    _, f = _single_domain(nvel, ntotal, nsteps)
    assert np.array_equal(got, interior(f, 1))


def test_cartesian_options_are_checked():
    with pytest.raises(ludwig_amd.LbmiError):       # direction 0 has one rank
        ludwig_amd.x_schedule(19, (4, 5, 6), 1, 4, 0, cartgrid=(1, 2, 2),
                              cartcoords=(0, 0, 0), dim=0)
    ops = ludwig_amd.x_schedule(19, (4, 5, 6), 1, 4, 3, cartgrid=(1, 2, 2),
                                cartcoords=(0, 1, 1), dim=1)
    # rank (0, 1, 1) = 3; below and above along Y: (0, 0, 1) = 1
    assert {op["peer"] for op in ops} == {1}
    ops = ludwig_amd.x_schedule(19, (4, 5, 6), 1, 4, 3, cartgrid=(1, 2, 2),
                                cartcoords=(0, 1, 1), dim=2)
    assert {op["peer"] for op in ops} == {2}


def test_zero_copy_messages_are_for_x_slabs_only():
    with pytest.raises(ludwig_amd.LbmiError):
        ludwig_amd.x_schedule(19, (4, 5, 6), 1, 2, 0, packed=False, cartdim=2)


def test_a_wrong_order_is_noticed():
    """Two ranks, one of them issuing its two sends in the other order: the
    planes land in the wrong halos and the result is not the single domain's.
    (What makes the test above a test of the product's ordering.)"""
    world, nvel, ntotal, nsteps = 2, 19, (8, 5, 6), 2
    res = _run(world, nvel, ntotal, nsteps, True, True, tamper="swap_sends")
    _, f = _single_domain(nvel, ntotal, nsteps)
    got = np.concatenate([res[r][0] for r in range(world)], axis=1)
    assert not np.array_equal(got, interior(f, 1))


def test_schedule_pairs_up_for_any_ring_size():
    """Host-only: for 1 .. 8 ranks every send of the schedule has exactly one
    receive on the peer that it meets under in-order matching, with the same
    length, and the buffers are the ones the layout says (sendhi -> the next
    rank's recvlo, sendlo -> the previous rank's recvhi)."""
    for nvel, cartdim in ((19, 0), (27, 0), (19, 1), (19, 2), (27, 2)):
        for world in range(1, 9):
            for packed in ((True, False) if cartdim == 0 else (True,)):
                sched = [ludwig_amd.x_schedule(nvel, (4, 5, 6), 1, world, r, packed=packed,
                                               cartdim=cartdim)
                         for r in range(world)]
                for r in range(world):
                    for peer in set(op["peer"] for op in sched[r]):
                        sends = [op for op in sched[r] if op["kind"] == "send" and op["peer"] == peer]
                        recvs = [op for op in sched[peer] if op["kind"] == "recv" and op["peer"] == r]
                        assert len(sends) == len(recvs)
                        for s, v in zip(sends, recvs):
                            assert s["count"] == v["count"]
                            if packed:
                                want = {"sendhi": "recvlo", "sendlo": "recvhi"}[s["buffer"]]
                                assert v["buffer"] == want
                            if packed and world > 2:
                                assert peer == ((r + 1) % world if s["buffer"] == "sendhi"
                                                else (r - 1) % world)


def test_schedule_pairs_up_on_grids_of_ranks():
    """Host-only: on grids of up to 3 x 3 x 3 ranks (LBMI_CART_GENERAL) every
    send of the schedule of a decomposed direction meets, in issue order, a
    receive of the same length on the peer, into the buffer the layout says;
    the peers are the Cartesian neighbours along that direction and nobody
    else; a direction with one rank has no schedule."""
    for nvel in (19, 27):
        for grid in ((2, 2, 1), (1, 2, 2), (2, 1, 3), (2, 2, 2), (3, 3, 1), (3, 2, 3), (3, 3, 3)):
            world = grid[0] * grid[1] * grid[2]
            decs = [ludwig_amd.CartDecomposition((6, 6, 6), grid, r) for r in range(world)]
            for dim in range(3):
                if grid[dim] < 2:
                    with pytest.raises(ludwig_amd.LbmiError):
                        ludwig_amd.x_schedule(nvel, decs[0].nlocal, 1, world, 0, cartgrid=grid,
                                              cartcoords=decs[0].coords, dim=dim)
                    continue
                sched = [ludwig_amd.x_schedule(nvel, d.nlocal, 1, world, d.rank, cartgrid=grid,
                                               cartcoords=d.coords, dim=dim) for d in decs]
                for r, d in enumerate(decs):
                    below, above = d.neighbours(dim)
                    assert {op["peer"] for op in sched[r]} == {below, above}
                    for peer in (below, above):
                        sends = [op for op in sched[r] if op["kind"] == "send" and op["peer"] == peer]
                        recvs = [op for op in sched[peer] if op["kind"] == "recv" and op["peer"] == r]
                        assert len(sends) == len(recvs) and sends
                        for s, v in zip(sends, recvs):
                            assert s["count"] == v["count"]
                            assert v["buffer"] == {"sendhi": "recvlo", "sendlo": "recvhi"}[s["buffer"]]
                            if grid[dim] > 2:
                                assert peer == (above if s["buffer"] == "sendhi" else below)
