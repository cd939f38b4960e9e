"""Row f4 on the device: flat walls and bounce-back on links (wall.c) through
the C-ABI against the compiled-reference fixtures and the oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                                    # noqa: E402
from tests.common import (golden_slip_names, golden_wall_names, interior,  # noqa: E402
                          load_golden, relmax)


def _setup(g, mode=0):
    import ludwig_amd
    import torch
    meta = g["meta"]
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), 1, mode=mode)
    lb.relaxation_set("m10", meta["eta"], meta["zeta"])
    st0 = np.zeros(lb.nall, dtype=np.int8)
    if meta["solid"] == 1:
        st0[2:4, 2:4, 2:4] = 1
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=st0)
    torch.cuda.synchronize()
    return lb, hy, meta


def _mark_colloids(lb, hy, g):
    """solid = 2 fixtures: the reference marked some fluid sites MAP_COLLOID
    AFTER its links were built; the bounce-back kernels test the map."""
    import torch
    hy.status.copy_(torch.tensor(g["status"], dtype=torch.int8))
    lb.wall_status_set(hy.status)
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", golden_wall_names())
def test_wall_map_and_links_exact(name):
    g = load_golden(name)
    lb, hy, meta = _setup(g)
    lb.wall_map(meta["isboundary"], hy.status)
    lb.synchronize()
    assert np.array_equal(hy.status.cpu().numpy(),
                          np.where(g["status"] == 2, 0, g["status"]))
    n = lb.wall_links_build(hy.status, meta["isboundary"])
    assert n == meta["nlink"]
    li, lj, lp, lu = lb.wall_links()
    assert np.array_equal(li, g["linki"]) and np.array_equal(lj, g["linkj"])
    assert np.array_equal(lp, g["linkp"]) and np.array_equal(lu, g["linku"])
    lb.free()


@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
@pytest.mark.parametrize("name", golden_wall_names())
def test_wall_steps_vs_reference(name, mode):
    """EAGER: three stages. FUSED_HALO: lb_collide and lb_halo leave the
    reference's state for the bounce-back; only the propagation is deferred
    into the next collision."""
    g = load_golden(name)
    lb, hy, meta = _setup(g, mode)
    lb.wall_map(meta["isboundary"], hy.status)
    lb.wall_links_build(hy.status, meta["isboundary"])
    lb.wall_velocity_set(meta["ubot"], meta["utop"])
    if meta["solid"] == 2:
        _mark_colloids(lb, hy, g)
    lb.lb_memcpy_h2d(g["f0"])
    nv = meta["nvel"]
    for n in range(meta["nsteps"]):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        if n == 0:
            f = lb.lb_memcpy_d2h().reshape(nv, -1)
            ref = g["f_bbl"].reshape(nv, -1)
            q = nv - g["linkp"]
            assert np.max(np.abs(f[q, g["linkj"]] - ref[q, g["linkj"]])) < 1e-15
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(f, 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-12
    fnet = lb.wall_momentum()
    scale = max(1.0, np.abs(np.array(meta["fnet"])).max())
    assert np.max(np.abs(fnet - np.array(meta["fnet"]))) < 1e-12 * scale
    assert np.array_equal(lb.wall_momentum(), np.zeros(3))     # read zeroes it
    lb.free()


@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
def test_couette_between_moving_walls(mode):
    """Walls at z = 0, Lz+1 moving with -/+ u_w along x: the flow relaxes to
    the linear Couette profile u_x(z) = u_w (2 z - Lz - 1)/Lz of half-way
    bounce-back (walls half a site outside the first/last fluid node), and
    mass is conserved to rounding."""
    import ludwig_amd
    import torch
    n = (4, 4, 16)
    uw = 0.01
    lb = ludwig_amd.LB(19, n, 1, mode=mode)
    lb.relaxation_set("bgk", 1.0 / 6.0, 1.0 / 6.0)      # tau = 1
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    torch.cuda.synchronize()
    lb.wall_map((0, 0, 1), hy.status)
    assert lb.wall_links_build(hy.status, (0, 0, 1)) == 2 * 4 * 4 * 5
    lb.wall_velocity_set((-uw, 0, 0), (uw, 0, 0))
    w = ludwig_amd.model(19)["wv"]
    f0 = np.zeros((19,) + lb.nall)
    for p in range(19):
        interior(f0[p], 1)[...] = w[p]
    lb.lb_memcpy_h2d(f0)
    for _ in range(3000):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        lb.lb_propagation()
    lb.lb_collide(hy)
    lb.synchronize()
    ux = interior(hy.u.cpu().numpy(), 1)[0].mean(axis=(0, 1))
    z = np.arange(1, n[2] + 1)
    exact = uw * (2.0 * z - n[2] - 1.0) / n[2]
    assert np.max(np.abs(ux - exact)) < 2e-6
    assert abs(lb.moments()[1] - n[0] * n[1] * n[2]) < 1e-9
    lb.free()


def test_wall_needs_eager_and_links():
    import ludwig_amd
    lb = ludwig_amd.LB(19, (4, 4, 4), 1, mode=ludwig_amd.FUSED)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_bbl()                                  # no links yet
    lb.wall_map((1, 0, 0), hy.status)
    lb.wall_links_build(hy.status, (1, 0, 0))
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_bbl()                                  # FUSED: no such state
    lb.free()
    lb = ludwig_amd.LB(19, (4, 4, 4), 1, mode=ludwig_amd.FUSED_HALO)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    lb.wall_map((1, 0, 0), hy.status)
    lb.wall_links_build(hy.status, (1, 0, 0))
    lb.lb_collide(hy)
    lb.lb_halo()
    lb.wall_bbl()                                      # the state exists
    lb.lb_propagation()
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_bbl()                                  # too late: deferred
    lb.free()


def test_wall_with_two_distributions():
    """ndist = 2: the order-parameter distribution bounces back on the same
    links (wall.c:1081-1088)."""
    import ludwig_amd
    import torch
    n = (6, 5, 4)
    lb = ludwig_amd.LB(19, n, 1, ndist=2)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    torch.cuda.synchronize()
    lb.wall_map((1, 0, 0), hy.status)
    lb.wall_links_build(hy.status, (1, 0, 0))
    lb.wall_velocity_set((0, 0.01, 0), (0, -0.02, 0))
    rng = np.random.default_rng(2)
    f0 = rng.random((38,) + lb.nall)
    lb.lb_memcpy_h2d(f0)
    lb.wall_bbl()
    f = lb.lb_memcpy_d2h().reshape(2, 19, -1)
    li, lj, lp, lu = lb.wall_links()
    m = ludwig_amd.model(19)
    uw = np.array([[0, 0, 0], [0, -0.02, 0], [0, 0.01, 0]])
    ref = f0.reshape(2, 19, -1).copy()
    for k in range(len(li)):
        cdotu = float(m["cv"][lp[k]] @ uw[lu[k]])
        for d in range(2):
            ref[d, 19 - lp[k], lj[k]] = f0.reshape(2, 19, -1)[d, lp[k], li[k]] \
                - 2.0 * 3.0 * m["wv"][lp[k]] * 1.0 * cdotu
    assert np.max(np.abs(f - ref)) < 1e-15
    lb.free()


@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
@pytest.mark.parametrize("bnd", [(1, 0, 0), (0, 0, 1), (1, 1, 1)])
def test_walls_on_the_slab_path(bnd, mode):
    """Walls with the X halo going through the RCCL ring (one rank = first
    and last slab): the periodic images that the exchange puts into the X
    halo planes are overridden by the bounce-back wherever a wall is."""
    import ludwig_amd
    import torch
    n = (8, 6, 7)
    p = lbo.make_param(19, n, 1, "m10", 0.1, 0.3)
    f0 = lbo.init_synthetic(p)
    out = []
    for ring in (False, True):
        lb = ludwig_amd.LB(19, n, 1, mode=mode)
        lb.relaxation_set("m10", 0.1, 0.3)
        if ring:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
        hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
        torch.cuda.synchronize()
        lb.wall_map(bnd, hy.status)
        lb.wall_links_build(hy.status, bnd)
        lb.wall_velocity_set((0, 0.01, 0), (0, -0.02, 0))
        lb.lb_memcpy_h2d(f0)
        for _ in range(5):
            lb.lb_collide(hy)
            lb.lb_halo()
            lb.wall_bbl()
            lb.lb_propagation()
        out.append((interior(lb.lb_memcpy_d2h(), 1).copy(), lb.wall_momentum()))
        lb.free()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.max(np.abs(out[0][1] - out[1][1])) < 1e-13


# Partial slip (wall_init_boundaries_slip, wall_bbl_slip_kernel)

@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
@pytest.mark.parametrize("name", golden_slip_names())
def test_slip_vs_reference(name, mode):
    """Links k, q, s identical to the reference's; the values its slip kernel
    wrote, four whole steps and the wall momentum."""
    g = load_golden(name)
    lb, hy, meta = _setup(g, mode)
    lb.wall_map(meta["isboundary"], hy.status)
    assert lb.wall_links_build(hy.status, meta["isboundary"]) == meta["nlink"]
    lb.wall_slip_set(hy.status, meta["sbot"], meta["stop"])
    lk, lq, ls = lb.wall_slip_links()
    assert np.array_equal(lk, g["linkk"])
    assert np.array_equal(lq, g["linkq"])
    assert np.array_equal(ls, g["links"])
    if meta["solid"] == 2:
        _mark_colloids(lb, hy, g)
    lb.lb_memcpy_h2d(g["f0"])
    nv = meta["nvel"]
    for n in range(meta["nsteps"]):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        if n == 0:
            f = lb.lb_memcpy_d2h().reshape(nv, -1)
            ref = g["f_bbl"].reshape(nv, -1)
            q = nv - g["linkp"]
            assert np.max(np.abs(f[q, g["linkj"]] - ref[q, g["linkj"]])) < 1e-15
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(f, 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-12
    fnet = lb.wall_momentum()
    scale = max(1.0, np.abs(np.array(meta["fnet"])).max())
    assert np.max(np.abs(fnet - np.array(meta["fnet"]))) < 1e-12 * scale
    # slip links pair up with one s: mass is conserved
    if meta["solid"] != 2:
        assert abs(interior(f, 1)[:, fl].sum() - interior(g["f0"], 1)[:, fl].sum()) < 1e-11
    lb.free()


def test_slip_arrays_entry_point_and_switch_off():
    """lbmi_wall_bbl_slip_arrays on caller-owned device arrays in the
    reference's types (int, int8_t, int8_t) = the handle's own links; all
    fractions zero = the no-slip kernel again; bad fractions are refused."""
    import ctypes
    import ludwig_amd
    import torch
    from ludwig_amd import lib as L
    g = load_golden("slip_q27_xy")
    lb, hy, meta = _setup(g)
    lb.wall_map(meta["isboundary"], hy.status)
    lb.wall_links_build(hy.status, meta["isboundary"])
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_slip_links()                            # not active yet
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_slip_set(hy.status, (1.5, 0, 0), (0, 0, 0))
    lb.wall_slip_set(hy.status, meta["sbot"], meta["stop"])
    rng = np.random.default_rng(5)
    f0 = rng.random((27,) + lb.nall)
    lb.lb_memcpy_h2d(f0)
    lb.wall_bbl()
    mine = lb.lb_memcpy_d2h()
    fnet_mine = lb.wall_momentum()

    _, stab = lbo.wall_slip_table(meta["sbot"], meta["stop"])
    dev = lb.device
    li, lj, lp = [torch.tensor(g[k], dtype=torch.int32, device=dev)
                  for k in ("linki", "linkj", "linkp")]
    lk = torch.tensor(g["linkk"], dtype=torch.int32, device=dev)
    lq = torch.tensor(g["linkq"].astype(np.int8), device=dev)
    ls = torch.tensor(g["links"].astype(np.int8), device=dev)
    fnet = torch.zeros(3, dtype=torch.float64, device=dev)
    lb.lb_memcpy_h2d(f0)
    torch.cuda.synchronize()
    L.check(L.library().lbmi_wall_bbl_slip_arrays(
        lb._h, len(g["linki"]), li.data_ptr(), lj.data_ptr(), lp.data_ptr(),
        lk.data_ptr(), lq.data_ptr(), ls.data_ptr(),
        (ctypes.c_double * 19)(*stab), fnet.data_ptr()))
    lb.synchronize()
    assert np.array_equal(lb.lb_memcpy_d2h(), mine)
    assert np.array_equal(fnet.cpu().numpy(), fnet_mine)

    # against the oracle on the same random state
    p = lbo.make_param(27, meta["nlocal"], 1, "m10", meta["eta"], meta["zeta"])
    ref = f0.copy()
    fo = np.zeros(3)
    lbo.wall_bbl_slip(p, ref, (g["linki"], g["linkj"], g["linkp"], g["linku"]),
                      (g["linkk"], g["linkq"], g["links"]), stab, fo)
    # (1-s) f_i + s f_k: one rounding (fma) here, two in the oracle
    assert np.max(np.abs(mine - ref)) < 2e-16
    assert np.max(np.abs(fnet_mine - fo)) < 1e-12 * max(1.0, np.abs(fo).max())

    # slip off: plain bounce-back
    lb.wall_slip_set(hy.status, (0, 0, 0), (0, 0, 0))
    lb.lb_memcpy_h2d(f0)
    lb.wall_bbl()
    f = lb.lb_memcpy_d2h().reshape(27, -1)
    q = 27 - g["linkp"]
    assert np.array_equal(f[q, g["linkj"]], f0.reshape(27, -1)[g["linkp"], g["linki"]])
    lb.free()


def test_slip_refuses_solid_blocks():
    """A link at a convex edge of a solid block has no wall normal: the
    reference asserts (wall.c:557-558); here the call fails."""
    import ludwig_amd
    g = load_golden("wall_q19_xyz")                     # walls + a solid block
    lb, hy, meta = _setup(g)
    lb.wall_map(meta["isboundary"], hy.status)
    lb.wall_links_build(hy.status, meta["isboundary"])
    with pytest.raises(ludwig_amd.LbmiError):
        lb.wall_slip_set(hy.status, (0.5, 0, 0), (0, 0, 0))
    lb.free()


@pytest.mark.parametrize("s", [0.0, 1.0])
def test_slip_channel_flow(s):
    """A body force along x between z walls: with free slip (s = 1) nothing
    holds the fluid and the profile stays flat (plug flow, u = F t / rho);
    with s = 0 the slip kernel is plain bounce-back and gives the parabola."""
    import ludwig_amd
    import torch
    n = (4, 4, 12)
    lb = ludwig_amd.LB(19, n, 1)
    lb.relaxation_set("bgk", 1.0 / 6.0, 1.0 / 6.0)
    fx = 1e-6
    lb.body_force_set((fx, 0, 0))
    hy = ludwig_amd.Hydro(lb.nall, lb.device, status=np.zeros(lb.nall, dtype=np.int8))
    torch.cuda.synchronize()
    lb.wall_map((0, 0, 1), hy.status)
    lb.wall_links_build(hy.status, (0, 0, 1))
    if s > 0.0:
        lb.wall_slip_set(hy.status, (0, 0, s), (0, 0, s))
    w = ludwig_amd.model(19)["wv"]
    f0 = np.zeros((19,) + lb.nall)
    for p in range(19):
        interior(f0[p], 1)[...] = w[p]
    lb.lb_memcpy_h2d(f0)
    nstep = 400
    for _ in range(nstep):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        lb.lb_propagation()
    lb.lb_collide(hy)
    lb.synchronize()
    ux = interior(hy.u.cpu().numpy(), 1)[0].mean(axis=(0, 1))
    if s == 1.0:
        assert np.max(np.abs(ux - ux.mean())) < 1e-12
        assert abs(ux.mean() - fx * (nstep + 0.5)) < 1e-3 * fx * nstep
    else:
        assert ux[0] < 0.5 * ux[n[2] // 2]             # held at the walls
        assert np.allclose(ux, ux[::-1], rtol=0, atol=1e-12)
    assert abs(lb.moments()[1] - n[0] * n[1] * n[2]) < 1e-9
    lb.free()


# --- links made by the caller: checked before a kernel can see them -------------

@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
@pytest.mark.parametrize("name", golden_wall_names())
def test_links_set_from_host_arrays(name, mode):
    """lbmi_wall_links_set with the reference's host link arrays (what the
    binding passes: wall->linki, linkj, linkp, linku) and the momentum into an
    accumulator the caller owns: the steps of the fixture."""
    import torch
    g = load_golden(name)
    lb, hy, meta = _setup(g, mode)
    lb.wall_map(meta["isboundary"], hy.status)
    lb.wall_links_set(g["linki"], g["linkj"], g["linkp"], g["linku"])
    assert lb.nlink == meta["nlink"]
    fnet = torch.zeros(3, dtype=torch.float64, device=hy.status.device)
    lb.wall_fnet_bind(fnet)
    lb.wall_velocity_set(meta["ubot"], meta["utop"])
    if meta["solid"] == 2:
        _mark_colloids(lb, hy, g)
    lb.lb_memcpy_h2d(g["f0"])
    for n in range(meta["nsteps"]):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.wall_bbl()
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(f, 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-12
    ref = np.array(meta["fnet"])
    assert np.max(np.abs(fnet.cpu().numpy() - ref)) < 1e-12 * max(1.0, np.abs(ref).max())
    assert np.all(lb.wall_momentum() == 0.0)        # nothing went to the handle's own
    lb.free()


def test_links_set_refuses_bad_records():
    """A record that is not a link of this lattice never reaches the device:
    p = 0 (the all-zero record of an array that was never filled), p = nvel,
    sites outside the array, j != i + c_p, a wall-velocity id out of range."""
    import ludwig_amd
    from ludwig_amd.lib import LbmiError
    g = load_golden("wall_q19_x")
    lb, hy, meta = _setup(g)
    li, lj, lp, lu = (g[k].copy() for k in ("linki", "linkj", "linkp", "linku"))
    nsite = int(np.prod(lb.nall))
    n = len(li) // 2
    for what, arr, val in (("p = 0", lp, 0), ("p = nvel", lp, meta["nvel"]),
                           ("i < 0", li, -1), ("i = nsite", li, nsite),
                           ("j elsewhere", lj, int(lj[n]) + 1), ("u = 3", lu, 3)):
        keep = arr[n]
        arr[n] = val
        with pytest.raises(LbmiError, match="link %d" % n):
            lb.wall_links_set(li, lj, lp, lu)
        arr[n] = keep
    lb.wall_links_set(np.zeros(0, np.int32), np.zeros(0, np.int32),
                      np.zeros(0, np.int32), np.zeros(0, np.int32))
    lb.wall_links_set(li, lj, lp, lu)                 # and the intact set is taken
    assert lb.nlink == len(li)
    lb.free()


def test_foreign_device_arrays_are_range_checked():
    """lbmi_wall_bbl_arrays on device arrays the caller owns: the kernel skips
    a record that would address outside f -- the all-zero record would touch
    f[nsite*nvel], one element past the end -- and the call that first sees
    the arrays reports it; the other links are bounced as usual."""
    import torch
    from ludwig_amd.lib import LbmiError
    g = load_golden("wall_q19_x")
    lb, hy, meta = _setup(g)
    nv = meta["nvel"]
    lb.wall_map(meta["isboundary"], hy.status)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_collide(hy)
    lb.lb_halo()
    dev = hy.status.device
    bad = len(g["linki"]) // 3
    arrs = []
    for k in ("linki", "linkj", "linkp", "linku"):
        a = g[k].astype(np.int32).copy()
        a[bad] = 0
        arrs.append(torch.tensor(a, device=dev))
    fnet = torch.zeros(3, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with pytest.raises(LbmiError, match="link record %d" % bad):
        lb.wall_bbl_arrays(*arrs, meta["ubot"], meta["utop"], fnet)
    f = lb.lb_memcpy_d2h().reshape(nv, -1)
    ref = g["f_bbl"].reshape(nv, -1)
    ok = np.arange(len(g["linki"])) != bad
    q, j = (nv - g["linkp"])[ok], g["linkj"][ok]
    assert np.max(np.abs(f[q, j] - ref[q, j])) < 1e-15
    # the same arrays again, repaired in place: accepted without a new check
    arrs[2][bad] = int(g["linkp"][bad])
    arrs[0][bad] = int(g["linki"][bad])
    arrs[1][bad] = int(g["linkj"][bad])
    torch.cuda.synchronize()
    lb.wall_bbl_arrays(*arrs, meta["ubot"], meta["utop"], fnet)
    lb.synchronize()
    lb.free()


def test_mode_set_at_every_call_point():
    """lbmi_lb_mode_set between any two calls of a step: the run continues in
    the new mode from the state the reference holds there (what the binding
    does when a run in LBMI_MODE=fused turns out to have wall links)."""
    import ludwig_amd
    nvel, nlocal = 19, (6, 5, 7)
    p = lbo.make_param(nvel, nlocal, 1, "m10", 0.1, 0.3, 1.0, (1e-6, 2e-6, 0.0))
    f0 = lbo.init_synthetic(p)
    f, fp = f0.copy(), np.zeros_like(f0)
    for _ in range(4):
        f, fp = lbo.step(p, f, fp)
    for point in range(3):
        for first, then in ((1, 3), (3, 0), (2, 1), (0, 1)):
            lb = ludwig_amd.LB(nvel, nlocal, 1, mode=first)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set((1e-6, 2e-6, 0.0))
            hy = ludwig_amd.Hydro(lb.nall, lb.device)
            lb.lb_memcpy_h2d(f0)
            for n in range(4):
                lb.lb_collide(hy)
                if n == 2 and point == 0:
                    lb.mode_set(then)
                lb.lb_halo()
                if n == 2 and point == 1:
                    lb.mode_set(then)
                lb.lb_propagation()
                if n == 2 and point == 2:
                    lb.mode_set(then)
            out = lb.lb_memcpy_d2h()
            assert relmax(interior(out, 1), interior(f, 1)) < 1e-12, (point, first, then)
            lb.free()


@pytest.mark.parametrize("nvel,nlocal,nhalo,scheme", [
    (19, (10, 7, 12), 1, "m10"), (19, (6, 8, 5), 2, "trt"), (27, (7, 6, 9), 1, "bgk"),
    (19, (9, 8, 1), 1, "m10"),          # a quasi-two-dimensional box
])
def test_halo_computed_by_the_collision_kernel_is_the_swapped_halo(nvel, nlocal, nhalo, scheme):
    """LBMI_MODE_FUSED_HALO on one rank: the kernel of lb_collide computes the
    width-1 halo shell of its own result (each shell site = the collision of
    its periodic image, k_propagate_collide_halo), and the lb_halo that
    follows has nothing left to do. What is observable between lb_halo and
    lb_propagation -- the array other Ludwig kernels act on -- must be what
    three halo copies deliver: compared BIT FOR BIT, interior and shell, with
    the same handle run with lbmi_tune halo_fold 0, with a force field and
    solid sites, step after step; then with a writer between lb_collide and
    lb_halo that owns up (lbmi_lb_dirty), where the halo swap must be real."""
    import ludwig_amd
    import torch
    from oracle import lb_oracle as lbo
    zeta = 0.3 if scheme == "m10" else 0.1
    p = lbo.make_param(nvel, nlocal, nhalo, scheme, 0.1, zeta, 1.0, (1e-6, 2e-6, -1e-6))
    f0 = lbo.init_synthetic(p)
    nall = lbo.nall(p)
    rng = np.random.default_rng(21)
    force = 1e-6 * rng.standard_normal((3,) + nall)
    status = np.zeros(nall, dtype=np.int8)
    if min(nlocal) > 4:
        status[nhalo + 2:nhalo + 4, nhalo + 1:nhalo + 3, nhalo + 2:nhalo + 5] = 1
    h = nhalo - 1
    shell = (slice(None),) + ((slice(h, -h) if h else slice(None)),) * 3
    seen = {}
    for fold in (0, 1):
        lb = ludwig_amd.LB(nvel, nlocal, nhalo, mode=ludwig_amd.FUSED_HALO)
        lb.relaxation_set(scheme, 0.1, zeta)
        lb.body_force_set((1e-6, 2e-6, -1e-6))
        lb.tune("halo_fold", fold)
        hy = ludwig_amd.Hydro(lb.nall, lb.device, force=force, status=status)
        lb.lb_memcpy_h2d(f0)
        rec = []
        for n in range(5):
            lb.lb_collide(hy)
            if n == 3:
                # somebody rewrites an interior plane between collide and halo
                # (the Lees-Edwards reprojection does, through lb_memcpy) and
                # says so: this halo swap has to be done
                lb.synchronize()
                lb.f[:, nhalo, :, :] *= 1.0 + 1e-3
                torch.cuda.synchronize()
                lb.lb_dirty()
            lb.lb_halo()
            lb.synchronize()
            torch.cuda.synchronize()
            rec.append(lb.f.cpu().numpy()[shell].copy())
            lb.lb_propagation()
        rec.append(interior(lb.lb_memcpy_d2h(), nhalo).copy())
        lb.synchronize()
        rec.append(interior(hy.u.cpu().numpy(), nhalo).copy())
        lb.free()
        seen[fold] = rec
    for a, b in zip(seen[0], seen[1]):
        assert np.array_equal(a, b)
