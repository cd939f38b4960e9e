"""N ranks of the slab decomposition on ONE GPU: an in-process ring of N
handles (lbmi_ring_t, one thread per rank) runs the product's multi-GPU step
-- interior launch, pack / exchange schedule / boundary launch against the
exchange buffers, three streams per rank -- with device-to-device copies in
place of ncclSend / ncclRecv (RCCL refuses two ranks on one device). The
schedule executed is lbmi_x_ops, the one RCCL gets (tests/test_slab_gloo.py
drives the same list over gloo on the CPU). Compared with the single-domain
oracle, including the two-rank case where previous and next rank coincide."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                       # noqa: E402
from tests.common import interior, relmax                  # noqa: E402

FBODY = (2e-6, -1e-6, 1e-6)


def _oracle(nvel, ntotal, nsteps, scheme="m10", force=None):
    p = lbo.make_param(nvel, ntotal, 1, scheme, 0.1, 0.3 if scheme == "m10" else 0.1,
                       1.0, FBODY)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    rho = np.zeros(lbo.nall(p))
    u = np.zeros((3,) + tuple(lbo.nall(p)))
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp, force=force, rho=rho, u=u)
    return p, f, rho, u


def _run_ring(world, nvel, ntotal, nsteps, mode, tune=(), scheme="m10",
              force=None, lazy=False, observe=None, dim=0, mid=None):
    """Each rank in a thread of its own. Returns per rank the interior f, rho,
    u, the local moments, what `observe(lb, hy, rank, step)` returned, and the
    ring size the handle reported."""
    import ludwig_amd
    import torch
    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, 1, dim=dim)
            lb = ludwig_amd.LB(nvel, dec.nlocal, 1, mode=mode, cartsz=world,
                               cartrank=rank, own_stream=True, cartdim=dim,
                               halo_scheme=ludwig_amd.HALO_REDUCED)
            lb.relaxation_set(scheme, 0.1, 0.3 if scheme == "m10" else 0.1)
            lb.body_force_set(FBODY)
            for k, v in tune:
                lb.tune(k, v)
            if lazy:
                lb.tune("hydro_lazy", 1)
            lb.comm_init_ring(ring)
            p = lbo.make_param(nvel, dec.nlocal, 1, scheme, 0.1, 0.3, 1.0, FBODY)
            f0 = lbo.init_synthetic(p, ntotal, dec.noffset)
            fl = None
            if force is not None:
                sl = [slice(None)] * 4
                sl[1 + dim] = slice(dec.noffset[dim], dec.noffset[dim] + dec.nlocal[dim] + 2)
                fl = np.ascontiguousarray(force[tuple(sl)])
            hy = ludwig_amd.Hydro(lb.nall, lb.device, force=fl)
            lb.lb_memcpy_h2d(f0)
            seen = []
            start.wait()
            for n in range(nsteps):
                lb.lb_collide(hy)
                if mid is not None:
                    mid(lb, hy, rank, n)             # between lb_collide and lb_halo
                lb.lb_halo()
                lb.lb_propagation()
                if observe is not None:
                    seen.append(observe(lb, hy, rank, n))
            if lazy:
                lb.hydro_sync()
            mo = lb.moments()
            f = lb.lb_memcpy_d2h()
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (interior(f, 1).copy(), interior(hy.rho.cpu().numpy(), 1),
                         interior(hy.u.cpu().numpy(), 1), mo, seen, lb.comm_info())
            start.wait()                 # nobody frees while a peer still reads
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    assert all(not t.is_alive() for t in threads)
    ring.free()                      # (refuses while a handle is attached)
    return out


def _join(out, k, dim=0):
    return np.concatenate([o[k] for o in out], axis=-3 + dim)


@pytest.mark.parametrize("tune", [
    (),                                              # the defaults: direct, packed, concurrent, blocked
    (("x_direct", 0),),                              # pack, messages, unpack
    (("x_direct", 0), ("x_packed", 0), ("blocked", 0)),   # zero-copy planes
    (("x_concurrent", 0),),
    (("blocked", 0),),
], ids=["direct", "unpack", "zerocopy", "serial_boundary", "soa"])
@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("nvel", [19, 27])
def test_fused_slabs_equal_single_domain(nvel, world, tune):
    ntotal = (12, 14, 14)            # slabs of 6, 4, 3 planes; 16 x 16 = 256 sites per plane
    nsteps = 5
    import ludwig_amd
    out = _run_ring(world, nvel, ntotal, nsteps, ludwig_amd.FUSED, tune)
    p, f, rho, u = _oracle(nvel, ntotal, nsteps)
    assert relmax(_join(out, 0), interior(f, 1)) < 1e-12
    assert relmax(_join(out, 1), interior(rho, 1)) < 1e-12
    assert relmax(_join(out, 2), interior(u, 1)) < 1e-12
    mref = lbo.moments(p, f)
    assert abs(sum(o[3][1] for o in out) - mref[1]) / mref[1] < 1e-12
    for o in out:
        assert o[5][0] == world and o[5][2] == 2


@pytest.mark.parametrize("mode_name", ["eager", "fused_halo"])
@pytest.mark.parametrize("world", [2, 3])
def test_staged_modes_on_slabs(world, mode_name):
    """lb_halo as a call of its own on slabs: X through the ring with the
    full or reduced selection, then Y and Z."""
    import ludwig_amd
    mode = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO}[mode_name]
    ntotal, nsteps = (6, 7, 5), 4
    out = _run_ring(world, 19, ntotal, nsteps, mode, scheme="bgk")
    p, f, rho, u = _oracle(19, ntotal, nsteps, scheme="bgk")
    assert relmax(_join(out, 0), interior(f, 1)) < 1e-12
    assert relmax(_join(out, 2), interior(u, 1)) < 1e-12


def test_one_plane_per_slab_and_thin_slabs():
    """nlocal[X] = 1 (first and last interior plane are the same plane, no
    interior launch) and 2 (no interior launch either)."""
    import ludwig_amd
    for world, ntotal in ((4, (4, 14, 14)), (3, (6, 6, 5))):
        out = _run_ring(world, 19, ntotal, 4, ludwig_amd.FUSED)
        p, f, rho, u = _oracle(19, ntotal, 4)
        assert relmax(_join(out, 0), interior(f, 1)) < 1e-12


def test_lazy_hydro_and_force_field_on_slabs():
    import ludwig_amd
    ntotal, nsteps, world = (8, 6, 7), 4, 2
    rng = np.random.default_rng(5)
    force = 1e-4 * (rng.random((3, ntotal[0] + 2, ntotal[1] + 2, ntotal[2] + 2)) - 0.5)
    out = _run_ring(world, 19, ntotal, nsteps, ludwig_amd.FUSED, force=force, lazy=True)
    p, f, rho, u = _oracle(19, ntotal, nsteps, force=force)
    assert relmax(_join(out, 0), interior(f, 1)) < 1e-12
    assert relmax(_join(out, 1), interior(rho, 1)) < 1e-12
    assert relmax(_join(out, 2), interior(u, 1)) < 1e-12


def test_observers_between_steps_do_not_disturb_the_exchange():
    """A flush between two steps (moments, a copy out) invalidates what the
    boundary launch left in the send buffers; a field halo in between uses
    other buffers. Every observation equals the single domain's at that step."""
    import ludwig_amd
    import torch
    ntotal, nsteps, world = (9, 6, 6), 6, 3
    refs = {}
    p = lbo.make_param(19, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    for n in range(nsteps):
        f, fp = lbo.step(p, f, fp)
        refs[n] = interior(f, 1).copy()

    def observe(lb, hy, rank, n):
        if n == 1:
            return ("f", lb.lb_memcpy_d2h())            # flush
        if n == 2:
            phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
            torch.cuda.synchronize()
            lb.field_halo_n(phi, 1)                       # another exchange in between
            lb.synchronize()
            return None
        if n == 4:
            return ("m", lb.moments())                   # flush again
        return None

    out = _run_ring(world, 19, ntotal, nsteps, ludwig_amd.FUSED, observe=observe)
    got1 = np.concatenate([interior(o[4][1][1], 1) for o in out], axis=1)
    assert relmax(got1, refs[1]) < 1e-12
    assert relmax(_join(out, 0), refs[nsteps - 1]) < 1e-12


def test_ring_needs_own_streams_and_matching_size():
    import ludwig_amd
    from ludwig_amd.lib import LbmiError
    ring = ludwig_amd.Ring(2)
    lb = ludwig_amd.LB(19, (4, 6, 6), 1, mode=ludwig_amd.FUSED, cartsz=2, cartrank=0)
    with pytest.raises(LbmiError, match="own stream"):
        lb.comm_init_ring(ring)
    lb.free()
    lb = ludwig_amd.LB(19, (4, 6, 6), 1, mode=ludwig_amd.FUSED, cartsz=3, cartrank=0,
                       own_stream=True)
    with pytest.raises(LbmiError, match="ring of 2"):
        lb.comm_init_ring(ring)
    lb.free()
    ring.free()


@pytest.mark.parametrize("world", [2, 3])
def test_binary_fluid_step_on_slabs(world):
    """BASELINE config 4 on slabs (nhalo 2): field halo of phi (two layers) and
    of u over the ring, the one-pass force + Cahn-Hilliard kernel, the FUSED LB
    step with its own exchange buffers in between -- against the oracle's
    single domain."""
    import ludwig_amd
    import torch
    ntotal, h, nsteps = (12, 10, 8), 2, 4
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(19, ntotal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(23)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(ntotal)
    f0 = lbo.init_synthetic(p)
    phi, f = phi0.copy(), f0.copy()
    fp = np.zeros_like(f)
    u = np.zeros((3,) + phi.shape)
    rho = np.zeros(phi.shape)
    for _ in range(nsteps):
        force = np.zeros((3,) + phi.shape)
        lbo.field_halo(p, phi, 2)
        grad, delsq = lbo.grad(p, phi, 7)
        lbo.symm_force(p, a, b, kappa, phi, grad, delsq, force)
        lbo.field_halo(p, u, 1)
        lbo.cahn_hilliard(p, a, b, kappa, mob, phi, delsq, u, order=2)
        u[...] = 0.0
        f, fp = lbo.step(p, f, fp, force, None, rho, u)

    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, h)
            x0, nx = dec.noffset[0], dec.nlocal[0]
            lb = ludwig_amd.LB(19, dec.nlocal, h, mode=ludwig_amd.FUSED, cartsz=world,
                               cartrank=rank, own_stream=True,
                               halo_scheme=ludwig_amd.HALO_REDUCED)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.fe_scheme_set(7, 2)
            lb.comm_init_ring(ring)
            hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
            pa = torch.from_numpy(np.ascontiguousarray(phi0[x0:x0 + nx + 2 * h])).to(lb.device)
            pb = torch.zeros_like(pa)
            torch.cuda.synchronize()
            lb.lb_memcpy_h2d(np.ascontiguousarray(f0[:, x0:x0 + nx + 2 * h]))
            start.wait()
            for _ in range(nsteps):
                lb.field_halo_n(pa, 2)
                lb.field_halo_n(hy.u, 1)
                lb.symmetric_step(a, b, kappa, mob, pa, hy.u, hy.force, pb, accumulate=False)
                pa, pb = pb, pa
                lb.step(hy)
            fo = lb.lb_memcpy_d2h()
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (interior(fo, h).copy(), interior(pa.cpu().numpy(), h).copy())
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    ring.free()
    assert relmax(np.concatenate([o[1] for o in out], axis=0), interior(phi, h)) < 1e-12
    assert relmax(np.concatenate([o[0] for o in out], axis=1), interior(f, h)) < 1e-12


def test_long_run_on_the_ring():
    """Four ranks, 240 steps back to back (lbmi_lb_run in chunks, the FIFO
    slots of the ring reused many times over, the next step's messages issued
    a step ahead every time), conserved totals read in between: the single
    domain after the same number of steps."""
    import ludwig_amd
    import torch
    world, ntotal, nsteps = 4, (16, 14, 14), 240
    p, f, rho, u = _oracle(19, ntotal, nsteps)
    mref = lbo.moments(p, f)
    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, 1)
            lb = ludwig_amd.LB(19, dec.nlocal, 1, mode=ludwig_amd.FUSED, cartsz=world,
                               cartrank=rank, own_stream=True,
                               halo_scheme=ludwig_amd.HALO_REDUCED)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set(FBODY)
            lb.tune("hydro_lazy", 1)
            lb.comm_init_ring(ring)
            pp = lbo.make_param(19, dec.nlocal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
            hy = ludwig_amd.Hydro(lb.nall, lb.device)
            lb.lb_memcpy_h2d(lbo.init_synthetic(pp, ntotal, dec.noffset))
            start.wait()
            mass = []
            for chunk in range(6):
                lb.run(hy, nsteps // 6)
                mass.append(lb.moments()[1])          # a flush every 40 steps
            lb.hydro_sync()
            fo = lb.lb_memcpy_d2h()
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (interior(fo, 1).copy(), mass, interior(hy.u.cpu().numpy(), 1).copy())
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    ring.free()
    assert relmax(_join(out, 0), interior(f, 1)) < 1e-12
    assert relmax(_join(out, 2), interior(u, 1)) < 1e-12
    for k in range(6):
        total = sum(o[1][k] for o in out)
        assert abs(total - mref[1]) / mref[1] < 1e-12        # mass is conserved throughout


@pytest.mark.parametrize("world", [2, 3])
def test_bench_peer_transport_line(world):
    """`python bench.py --gpus N --transport peer`: the N-rank slab step in one
    process over the peer ring (one device here: the ranks share it; N devices
    on an N-GPU node, the planes then travel as hipMemcpyPeerAsync). The
    contract's JSON line, conserved mass, a step time per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world),
                        "--transport", "peer", "--size", "48", "32", "32", "--steps", "12",
                        "--warmup", "2", "--cpu-baseline", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == world and d["unit"] == "MLUPS" and d["value"] > 0
    assert d["config"]["transport"].startswith("peer")
    assert len(d["slab_step"]) == world
    assert all(p["transport"] == "peer" and p["step_ms"] > 0 for p in d["slab_step"])
    assert d["check"]["mass_drift_rel"] < 1e-12


@pytest.mark.parametrize("mode", ["eager", "fused_halo", "fused", "fused_unpack", "fused_lazy"])
@pytest.mark.parametrize("world,nvel,dim", [(2, 19, 2), (3, 19, 2), (2, 27, 2), (3, 19, 1), (2, 27, 1),
                                            (4, 19, 2)])
def test_slabs_along_y_or_z_equal_single_domain(world, nvel, dim, mode):
    """grid 1_1_N (BASELINE config 3 as written: z slabs) and 1_N_1: the
    planes of the decomposed direction are gathered into the message buffers
    and scattered into the halo (k_halo_pack_x / k_halo_unpack_x with a
    direction), the two other passes stay local, in the order X, Y, Z. FUSED
    on such slabs: one launch over all planes beside the messages, then the
    face launch for the two boundary planes against the exchange buffers
    (k_propagate_collide_face), the next step's messages right behind it.
    With a force field, rho and u checked."""
    import ludwig_amd
    modes = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO,
             "fused": ludwig_amd.FUSED, "fused_unpack": ludwig_amd.FUSED,
             "fused_lazy": ludwig_amd.FUSED}
    ntotal = [10, 6, 14]
    ntotal[dim] = 12                 # slabs of 6, 4, 3 planes (4 ranks: 3)
    ntotal = tuple(ntotal)
    nsteps = 5
    rng = np.random.default_rng(11)
    force = 1e-6 * rng.standard_normal((3,) + tuple(n + 2 for n in ntotal))
    out = _run_ring(world, nvel, ntotal, nsteps, modes[mode], force=force, dim=dim,
                    tune=(("x_direct", 0),) if mode == "fused_unpack" else (),
                    lazy=(mode == "fused_lazy"))
    p, f, rho, u = _oracle(nvel, ntotal, nsteps, force=force)
    assert relmax(_join(out, 0, dim), interior(f, 1)) < 1e-12
    assert relmax(_join(out, 1, dim), interior(rho, 1)) < 1e-12
    assert relmax(_join(out, 2, dim), interior(u, 1)) < 1e-12
    assert all(o[5][0] == world and o[5][2] == 2 for o in out)


@pytest.mark.parametrize("blocked", [0, 1], ids=["soa", "blocked"])
@pytest.mark.parametrize("world", [2, 3])
def test_f_rewritten_between_collide_and_halo_of_a_fused_slab_step(world, blocked):
    """A fused slab step sends the planes of its NEXT exchange ahead, right
    after its boundary launch. If the caller then rewrites f behind the
    library's back between lb_collide and lb_halo -- the reference's
    lb_le_apply_boundary_conditions does, through lb_memcpy both ways
    (model_le.c:72-83) -- those planes are stale although nothing is pending
    (the flush of the copy out has nothing to do and leaves them valid):
    lbmi_lb_dirty, which the binding calls after every host -> device copy,
    makes the next step pack and exchange afresh. Here every rank scales its
    FIRST interior plane -- a plane the lower neighbour pulls from -- after
    the collision of the fourth step; the single domain does the same."""
    import ludwig_amd
    import torch
    nvel, ntotal, nsteps = 19, (12, 14, 14), 7
    nl = ntotal[0] // world

    def mid(lb, hy, rank, n):
        if n == 3:
            lb.lb_flush()                    # what lb_memcpy (D2H) does first
            lb.synchronize()
            lb.f[:, 1, :, :] *= 1.0 + 1e-3
            torch.cuda.synchronize()
            lb.lb_dirty()

    # (SoA: the flush has nothing to do and the send buffers stay "valid";
    # blocked: the flush converts the order, which invalidates them by itself)
    out = _run_ring(world, nvel, ntotal, nsteps, ludwig_amd.FUSED, mid=mid,
                    tune=(("blocked", blocked),))
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    for n in range(nsteps):
        lbo.collide(p, f)
        if n == 3:
            for r in range(world):
                f[:, 1 + r * nl, :, :] *= 1.0 + 1e-3
        lbo.halo(p, f)
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    assert relmax(_join(out, 0), interior(f, 1)) < 1e-12


@pytest.mark.parametrize("mode_name", ["eager", "fused_halo"])
@pytest.mark.parametrize("dim", [0, 2], ids=["x", "z"])
@pytest.mark.parametrize("bnd", [(1, 0, 0), (0, 1, 1), (1, 1, 1)], ids=["x", "yz", "xyz"])
@pytest.mark.parametrize("world", [2, 3])
def test_walls_on_slabs_equal_single_domain(world, bnd, dim, mode_name):
    """Flat moving walls (wall_init_map, wall_init_boundaries, wall_bbl:
    wall.c:399-451, 960-1107, 1219-1268) on a lattice cut into slabs: every
    rank marks and links ITS part -- in the decomposed direction only the
    first rank has the low wall and only the last one the high wall (noffset
    in wall.c:1236-1240) -- and bounces back between lb_halo and
    lb_propagation. The joined slabs are the single domain's distributions
    (whose wall steps are checked against the compiled reference's fixtures in
    test_gpu_wall.py) and the ranks' wall momenta add up to its momentum."""
    import ludwig_amd
    import torch
    mode = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO}[mode_name]
    nvel, ntotal, nsteps = 19, (12, 6, 12), 6
    ubot, utop = (0.0, 0.01, 0.002), (0.003, -0.02, 0.0)
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f0 = lbo.init_synthetic(p)

    def steps(lb, hy):
        for _ in range(nsteps):
            lb.lb_collide(hy)
            lb.lb_halo()
            lb.wall_bbl()
            lb.lb_propagation()

    def walls(lb):
        hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
        torch.cuda.synchronize()
        lb.wall_map(bnd, hy.status)
        n = lb.wall_links_build(hy.status, bnd)
        lb.wall_velocity_set(ubot, utop)
        return hy, n

    lb = ludwig_amd.LB(nvel, ntotal, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    hy, nlink = walls(lb)
    lb.lb_memcpy_h2d(f0)
    steps(lb, hy)
    ref = interior(lb.lb_memcpy_d2h(), 1).copy()
    fnet = lb.wall_momentum()
    lb.free()

    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, 1, dim=dim)
            lb = ludwig_amd.LB(nvel, dec.nlocal, 1, mode=mode, cartsz=world,
                               cartrank=rank, own_stream=True, cartdim=dim)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set(FBODY)
            lb.comm_init_ring(ring)
            hy, n = walls(lb)
            sl = [slice(None)] * 4
            sl[1 + dim] = slice(dec.noffset[dim], dec.noffset[dim] + dec.nlocal[dim] + 2)
            lb.lb_memcpy_h2d(np.ascontiguousarray(f0[tuple(sl)]))
            start.wait()
            steps(lb, hy)
            f = interior(lb.lb_memcpy_d2h(), 1).copy()
            out[rank] = (f, lb.wall_momentum(), n)
            lb.synchronize()
            torch.cuda.synchronize()
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    assert all(not t.is_alive() for t in threads)
    ring.free()
    assert sum(o[2] for o in out) == nlink
    got = _join(out, 0, dim)
    # (sites next to a wall in the decomposed direction included)
    assert relmax(got, ref) < 1e-13
    assert np.max(np.abs(sum(o[1] for o in out) - fnet)) < 1e-12 * max(1.0, np.abs(fnet).max())


def _run_cart(grid, nvel, ntotal, nsteps, mode, scheme_halo, walls=None):
    """One thread per rank of a Cartesian grid of handles on one device
    (LBMI_CART_GENERAL). Returns per rank (offset, interior f, rho, u, wall
    momentum, links)."""
    import ludwig_amd
    import torch
    world = grid[0] * grid[1] * grid[2]
    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)
    p0 = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f0 = lbo.init_synthetic(p0)

    def rank_main(rank):
        try:
            dec = ludwig_amd.CartDecomposition(ntotal, grid, rank, 1)
            lb = ludwig_amd.LB(nvel, dec.nlocal, 1, mode=mode, own_stream=True,
                               cartgrid=grid, cartcoords=dec.coords,
                               halo_scheme=scheme_halo)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set(FBODY)
            lb.comm_init_ring(ring)
            status = np.zeros(lb.nall, dtype=np.int8) if walls else None
            hy = ludwig_amd.Hydro(lb.nall, lb.device, status=status)
            nlink = 0
            if walls:
                torch.cuda.synchronize()
                lb.wall_map(walls[0], hy.status)
                nlink = lb.wall_links_build(hy.status, walls[0])
                lb.wall_velocity_set(walls[1], walls[2])
            lb.lb_memcpy_h2d(np.ascontiguousarray(f0[(slice(None),) + dec.local_block()]))
            start.wait()
            for _ in range(nsteps):
                lb.lb_collide(hy)
                lb.lb_halo()
                if walls:
                    lb.wall_bbl()
                lb.lb_propagation()
            f = interior(lb.lb_memcpy_d2h(), 1).copy()
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (dec.noffset, f, interior(hy.rho.cpu().numpy(), 1),
                         interior(hy.u.cpu().numpy(), 1),
                         lb.wall_momentum() if walls else None, nlink, lb.state())
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    assert all(not t.is_alive() for t in threads)
    ring.free()
    return out


def _assemble(out, k, shape):
    a = np.zeros(shape)
    for o in out:
        off, blk = o[0], o[k]
        a[..., off[0]:off[0] + blk.shape[-3], off[1]:off[1] + blk.shape[-2],
          off[2]:off[2] + blk.shape[-1]] = blk
    return a


@pytest.mark.parametrize("mode_name", ["eager", "fused_halo", "fused"])
@pytest.mark.parametrize("grid,nvel,halo", [
    ((2, 2, 1), 19, "full"), ((1, 2, 2), 19, "reduced"), ((2, 1, 2), 27, "reduced"),
    ((2, 2, 2), 19, "full"), ((2, 2, 2), 27, "reduced"), ((3, 2, 1), 19, "full"),
])
def test_cartesian_decomposition_equals_single_domain(grid, nvel, halo, mode_name):
    """More than one direction decomposed (lbmi_options_t::cartdim =
    LBMI_CART_GENERAL; the reference's default grid for N ranks is
    MPI_Dims_create's, 2_2_2 for eight): the halo swap is the reference's
    sequence of passes X, Y, Z (halo_swap.c:709-1063), a decomposed direction
    over the transport, the others wrapped on the rank, edges and corners
    completing through the halos of the earlier passes. `fused` runs as
    `halo` on such a grid. Against the single-domain oracle."""
    import ludwig_amd
    mode = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO,
            "fused": ludwig_amd.FUSED}[mode_name]
    sch = ludwig_amd.HALO_FULL if halo == "full" else ludwig_amd.HALO_REDUCED
    ntotal, nsteps = (4 * grid[0], 3 * grid[1], 4 * grid[2]), 5
    out = _run_cart(grid, nvel, ntotal, nsteps, mode, sch)
    p, f, rho, u = _oracle(nvel, ntotal, nsteps)
    assert relmax(_assemble(out, 1, (nvel,) + ntotal), interior(f, 1)) < 1e-12
    assert relmax(_assemble(out, 2, ntotal), interior(rho, 1)) < 1e-12
    assert relmax(_assemble(out, 3, (3,) + ntotal), interior(u, 1)) < 1e-12


@pytest.mark.parametrize("mode_name", ["eager", "fused_halo"])
@pytest.mark.parametrize("bnd", [(1, 0, 0), (0, 1, 1), (1, 1, 1)], ids=["x", "yz", "xyz"])
def test_walls_on_a_cartesian_decomposition(bnd, mode_name):
    """Walls on a 2 x 2 x 1 and a 1 x 2 x 2 grid of ranks: in every decomposed
    direction only the first rank has the low wall and only the last one the
    high wall; joined, the blocks are the single domain's distributions and
    the ranks' wall momenta add up to its momentum."""
    import ludwig_amd
    import torch
    mode = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO}[mode_name]
    nvel, ntotal, nsteps = 19, (8, 6, 8), 5
    ubot, utop = (0.0, 0.01, 0.002), (0.003, -0.02, 0.0)
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    lb = ludwig_amd.LB(nvel, ntotal, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    torch.cuda.synchronize()
    lb.wall_map(bnd, hy.status)
    nlink = lb.wall_links_build(hy.status, bnd)
    lb.wall_velocity_set(ubot, utop)
    lb.lb_memcpy_h2d(lbo.init_synthetic(p))
    for _ in range(nsteps):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        lb.lb_propagation()
    ref = interior(lb.lb_memcpy_d2h(), 1).copy()
    fnet = lb.wall_momentum()
    lb.free()
    for grid in ((2, 2, 1), (1, 2, 2)):
        out = _run_cart(grid, nvel, ntotal, nsteps, mode, ludwig_amd.HALO_FULL,
                        walls=(bnd, ubot, utop))
        assert sum(o[5] for o in out) == nlink
        assert relmax(_assemble(out, 1, (nvel,) + ntotal), ref) < 1e-13
        tot = sum(o[4] for o in out)
        assert np.max(np.abs(tot - fnet)) < 1e-12 * max(1.0, np.abs(fnet).max())


@pytest.mark.parametrize("grid", [(2, 2, 1), (1, 2, 2), (2, 2, 2)])
def test_field_halo_of_two_layers_on_a_cartesian_decomposition(grid):
    """field_halo of a three-component field with a halo two sites wide (phi
    of the free-energy runs, hydro->u) on a grid of ranks: every layer of
    every decomposed direction as its own packed plane over the transport,
    the passes in the order X, Y, Z -- afterwards every halo site, edges and
    corners included, holds the value of its periodic image in the global
    lattice."""
    import ludwig_amd
    import torch
    nh = 2
    ntotal = (4 * grid[0], 4 * grid[1], 4 * grid[2])
    world = grid[0] * grid[1] * grid[2]
    # a global field whose value tells component and site apart
    x, y, z = np.meshgrid(*[np.arange(n) for n in ntotal], indexing="ij")
    glob = np.stack([(c + 1) * 1000000.0 + (x * ntotal[1] + y) * ntotal[2] + z for c in range(3)])
    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.CartDecomposition(ntotal, grid, rank, nh)
            lb = ludwig_amd.LB(19, dec.nlocal, nh, mode=ludwig_amd.FUSED_HALO, own_stream=True,
                               cartgrid=grid, cartcoords=dec.coords)
            lb.comm_init_ring(ring)
            a = np.full((3,) + lb.nall, -1.0)
            sl = tuple(slice(dec.noffset[d], dec.noffset[d] + dec.nlocal[d]) for d in range(3))
            interior(a, nh)[...] = glob[(slice(None),) + sl]
            t = torch.from_numpy(a).to(lb.device)
            torch.cuda.synchronize()
            start.wait()
            lb.field_halo_n(t, nh)
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (dec.noffset, dec.nlocal, t.cpu().numpy())
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    ring.free()
    for off, nl, a in out:
        idx = [np.arange(off[d] - nh, off[d] + nl[d] + nh) % ntotal[d] for d in range(3)]
        want = glob[:, idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]]
        assert np.array_equal(a, want)


@pytest.mark.parametrize("mode_name", ["eager", "fused_halo", "fused"])
@pytest.mark.parametrize("dim", [0, 2], ids=["x", "z"])
@pytest.mark.parametrize("world", [2, 3])
def test_relaxed_stress_collision_on_slabs(world, dim, mode_name):
    """lb_collide with fe->use_stress_relaxation (lbmi_lb_collide_fe) on slabs:
    in `halo` the pending propagation runs inside the collision, pulling from
    the halo planes the exchange has filled (k_propagate_collide_fe without
    index wrap); in `fused` on slabs the halo swap that mode had only noted is
    done first, then the same; the
    joined slabs are the single domain's distributions, rho and u (whose steps
    are checked against the compiled reference's fixtures in
    test_gpu_binary.py)."""
    import ludwig_amd
    import torch
    mode = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO,
            "fused": ludwig_amd.FUSED}[mode_name]
    nvel, ntotal, nsteps = 19, (12, 6, 12), 5
    a, b, kappa = -0.00625, 0.00625, 0.004
    rng = np.random.default_rng(21)
    nall = tuple(n + 2 for n in ntotal)
    phi = 0.1 * rng.standard_normal(nall)
    grad = 0.01 * rng.standard_normal((3,) + nall)
    delsq = 0.01 * rng.standard_normal(nall)
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f0 = lbo.init_synthetic(p)

    def run(lb, sl):
        dev = lb.device
        hy = ludwig_amd.Hydro(lb.nall, dev)
        ph = torch.from_numpy(np.ascontiguousarray(phi[sl[1:]])).to(dev)
        gr = torch.from_numpy(np.ascontiguousarray(grad[sl])).to(dev)
        d2 = torch.from_numpy(np.ascontiguousarray(delsq[sl[1:]])).to(dev)
        lb.lb_memcpy_h2d(np.ascontiguousarray(f0[sl]))
        torch.cuda.synchronize()
        return hy, ph, gr, d2

    def steps(lb, hy, ph, gr, d2):
        for _ in range(nsteps):
            lb.lb_collide_fe(hy, a, b, kappa, ph, gr, d2)
            lb.lb_halo()
            lb.lb_propagation()

    lb = ludwig_amd.LB(nvel, ntotal, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    whole = (slice(None),) * 4
    hy, ph, gr, d2 = run(lb, whole)
    steps(lb, hy, ph, gr, d2)
    ref = (interior(lb.lb_memcpy_d2h(), 1).copy(), interior(hy.rho.cpu().numpy(), 1).copy(),
           interior(hy.u.cpu().numpy(), 1).copy())
    lb.free()

    ring = ludwig_amd.Ring(world)
    out = [None] * world
    err = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, 1, dim=dim)
            lb = ludwig_amd.LB(nvel, dec.nlocal, 1, mode=mode, cartsz=world, cartrank=rank,
                               own_stream=True, cartdim=dim)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set(FBODY)
            lb.comm_init_ring(ring)
            sl = [slice(None)] * 4
            sl[1 + dim] = slice(dec.noffset[dim], dec.noffset[dim] + dec.nlocal[dim] + 2)
            hy, ph, gr, d2 = run(lb, tuple(sl))
            start.wait()
            steps(lb, hy, ph, gr, d2)
            f = interior(lb.lb_memcpy_d2h(), 1).copy()
            lb.synchronize()
            torch.cuda.synchronize()
            out[rank] = (f, interior(hy.rho.cpu().numpy(), 1), interior(hy.u.cpu().numpy(), 1))
            start.wait()
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            start.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not err, err
    ring.free()
    for k in range(3):
        assert relmax(_join(out, k, dim), ref[k]) < 1e-13, k
