"""Parity of the HIP path (through the C-ABI, ludwig_amd.LB) with the oracle
and with the compiled-reference golden vectors. Needs an MI355X.

Tolerances: bit-exact for pure data movement (halo, propagation); for the
collision max|df|/max|f| <= 1e-12 and 1e-12 relative on conserved density and
momentum (BASELINE.json north_star). The collision cannot be
bit-exact: the reference sums dense 19x19 transforms without FMA on x86, the
kernel evaluates the same linear map in factored form with FMA contraction.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                       # noqa: E402
from tests.common import (RTOL_CONSERVED, RTOL_F, golden_names, interior,  # noqa: E402
                          load_golden, momentum_scale, relmax, shell1,
                          status_from_meta, xplanes)
from tests.regression_cases import (close_as_printed, initial_f,  # noqa: E402
                                    load_expected)


def make_lb(meta, mode=0, halo_scheme=0, **kw):
    import ludwig_amd
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), meta["nhalo"],
                       mode=mode, halo_scheme=halo_scheme, **kw)
    lb.relaxation_set(meta["scheme"], meta["eta"], meta["zeta"], meta["rho0"])
    lb.body_force_set(meta["fbody"])
    return lb


def make_hydro(lb, g, meta):
    import ludwig_amd
    st = status_from_meta(meta)
    return ludwig_amd.Hydro(lb.nall, lb.device, force=g["force"],
                            status=st if meta["solid"] else None)


def oracle_param(meta):
    return lbo.make_param(meta["nvel"], meta["nlocal"], meta["nhalo"],
                          meta["scheme"], meta["eta"], meta["zeta"],
                          meta["rho0"], meta["fbody"])


def dev(lb, a):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).to(lb.device)
    torch.cuda.synchronize(lb.device)
    return t


def host(lb, t):
    lb.synchronize()
    return t.cpu().numpy()


# --- golden vectors, one operator at a time (EAGER) --------------------------

@pytest.mark.parametrize("name", golden_names())
def test_collide_vs_reference(name):
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = make_lb(meta)
    hy = make_hydro(lb, g, meta)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_collide(hy)
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, h), interior(g["f_collide"], h)) < RTOL_F
    rho = host(lb, hy.rho)
    u = host(lb, hy.u)
    st = status_from_meta(meta)
    fluid = interior(st, h) == 0
    assert relmax(interior(rho, h)[fluid], interior(g["rho"], h)[fluid]) < RTOL_F
    assert relmax(interior(u, h)[:, fluid], interior(g["u"], h)[:, fluid]) < RTOL_F
    if meta["solid"]:
        # non-fluid sites untouched (collision.c:299-304)
        assert np.array_equal(interior(f, h)[:, ~fluid],
                              interior(g["f0"], h)[:, ~fluid])
    lb.free()


@pytest.mark.parametrize("name", [n for n in golden_names()
                                  if "f_halo" in load_golden(n)])
def test_halo_full_exact(name):
    g = load_golden(name)
    meta = g["meta"]
    lb = make_lb(meta)
    # start from the reference's own post-collision state (NaN -> 0 in the
    # never-exchanged outer layers, see tests/common.py:shell1)
    lb.lb_memcpy_h2d(np.nan_to_num(g["f_collide"]))
    lb.lb_halo()
    f = lb.lb_memcpy_d2h()
    h = meta["nhalo"]
    assert np.array_equal(shell1(f, h), shell1(g["f_halo"], h))
    lb.free()


@pytest.mark.parametrize("name", golden_names())
def test_halo_propagation_exact(name):
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = make_lb(meta)
    lb.lb_memcpy_h2d(np.nan_to_num(g["f_collide"]))
    lb.lb_halo()
    lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    assert np.array_equal(xplanes(f, h), xplanes(np.nan_to_num(g["f_prop"]), h))
    lb.free()


@pytest.mark.parametrize("name", golden_names())
def test_reduced_halo_same_interior(name):
    # test_model.c:546-647: only populations that re-enter are required
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = make_lb(meta, halo_scheme=2)
    lb.lb_memcpy_h2d(np.nan_to_num(g["f_collide"]))
    lb.lb_halo()
    lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    assert np.array_equal(interior(f, h), interior(g["f_prop"], h))
    lb.free()


# --- whole time steps: EAGER and FUSED against the reference -----------------

@pytest.mark.parametrize("mode", [0, 1, 2, 101, 3], ids=["eager", "fused", "inplace", "fused_soa", "fused_halo"])
@pytest.mark.parametrize("halo_scheme", [0, 2], ids=["full", "reduced"])
@pytest.mark.parametrize("name", golden_names())
def test_steps_vs_reference(name, mode, halo_scheme):
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = make_lb(meta, mode=mode, halo_scheme=halo_scheme)
    hy = make_hydro(lb, g, meta)
    lb.lb_memcpy_h2d(g["f0"])
    for _ in range(meta["nsteps"]):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.lb_propagation()
    st = hy.status
    mo = lb.moments(st)                       # flushes in FUSED mode
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, h), interior(g["f_final"], h)) < RTOL_F
    p = oracle_param(meta)
    mg = lbo.moments(p, np.ascontiguousarray(np.nan_to_num(g["f_final"])),
                     status_from_meta(meta))
    assert mo[0] == mg[0]
    assert abs(mo[1] - mg[1]) / mg[1] < RTOL_CONSERVED
    gscale = momentum_scale(g["f_final"], lbo.model(meta["nvel"])["cv"], h)
    assert np.max(np.abs(mo[5:8] - mg[5:8])) / gscale < RTOL_CONSERVED
    assert abs(mo[3] - mg[3]) < 1e-13 and abs(mo[4] - mg[4]) < 1e-13
    lb.free()


@pytest.mark.parametrize("name", ["q19_m10", "q19_trt_ffield", "q27_bgk"])
def test_fused_equals_eager_bitwise_inputs(name):
    """FUSED must reproduce EAGER: same collision arithmetic on the same
    pulled values. Agreement is checked at 1e-14 (identical in practice)."""
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    out = []
    for mode in (0, 1, 2, 101, 3):
        lb = make_lb(meta, mode=mode)
        hy = make_hydro(lb, g, meta)
        lb.lb_memcpy_h2d(g["f0"])
        for _ in range(7):
            lb.step(hy)
        out.append(interior(lb.lb_memcpy_d2h(), h).copy())
        lb.free()
    assert relmax(out[1], out[0]) < 1e-14
    assert relmax(out[2], out[0]) < 1e-14
    assert np.array_equal(out[3], out[1])       # blocked / SoA order: same arithmetic
    assert relmax(out[4], out[0]) < 1e-14       # FUSED_HALO


@pytest.mark.parametrize("mode", [1, 2, 101, 3], ids=["fused", "inplace", "fused_soa", "fused_halo"])
@pytest.mark.parametrize("name", ["q19_bgk_ffield", "q27_m10_ffield", "q19_m10_solid"])
def test_flush_at_every_call_point(name, mode):
    """A device-to-host copy (which flushes) placed after ANY call of ANY
    step must return what EAGER holds at that point, and must not disturb
    the run. Probe points: after lb_collide (0), lb_halo (1),
    lb_propagation (2) of steps 0..3."""
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    nsteps = 4

    def run(run_mode, probe):
        lb = make_lb(meta, mode=run_mode)
        hy = make_hydro(lb, g, meta)
        lb.lb_memcpy_h2d(g["f0"])
        seen = None
        for n in range(nsteps):
            for k, call in enumerate((lambda: lb.lb_collide(hy), lb.lb_halo,
                                      lb.lb_propagation)):
                call()
                if probe == (n, k):
                    seen = interior(lb.lb_memcpy_d2h(), h).copy()
        final = interior(lb.lb_memcpy_d2h(), h).copy()
        lb.free()
        return seen, final

    for n in range(nsteps):
        for k in range(3):
            ref_seen, ref_final = run(0, (n, k))
            seen, final = run(mode, (n, k))
            assert relmax(seen, ref_seen) < 1e-14, (n, k)
            assert relmax(final, ref_final) < 1e-14, (n, k)


def test_inplace_keeps_one_array():
    """INPLACE streams within f: the array behind lb.f does not change from
    step to step and fprime is never written by the step kernels."""
    import ludwig_amd
    g = load_golden("q19_m10")
    meta = g["meta"]
    lb = make_lb(meta, mode=ludwig_amd.INPLACE)
    lb.lb_memcpy_h2d(g["f0"])
    ptr0 = lb.f.data_ptr()
    lb.fprime.fill_(-7.0)
    import torch
    torch.cuda.synchronize()
    for _ in range(6):
        lb.step(None)
        assert lb.f.data_ptr() == ptr0
    lb.synchronize()
    assert float(lb.fprime.min()) == -7.0 and float(lb.fprime.max()) == -7.0
    lb.free()


def test_fused_state_errors():
    import ludwig_amd
    lb = ludwig_amd.LB(19, (4, 4, 4), 1, mode=1)
    lb.lb_halo()
    lb.lb_propagation()
    with pytest.raises(ludwig_amd.LbmiError):
        lb.lb_propagation()              # two propagations, no collision
    lb.lb_flush()
    lb.lb_collide(None)
    lb.free()


# --- stateless fused kernel with a real halo (wrap off) ----------------------

@pytest.mark.parametrize("name", ["q19_m10_fbody", "q19_bgk_ffield", "q27_m10_ffield",
                                  "q19_m10_nh2_ffield"])
def test_propagate_collide_with_halo(name):
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    p = oracle_param(meta)
    lb = make_lb(meta)
    hy = make_hydro(lb, g, meta)
    # oracle: halo, propagate, collide
    f = np.nan_to_num(g["f_collide"]).copy()
    lbo.halo(p, f)
    fp = np.zeros_like(f)
    lbo.propagate(p, f, fp)
    lbo.collide(p, fp, g["force"].copy(), status_from_meta(meta))
    # device: halo kernel then fused kernel without index wrap
    a = dev(lb, np.nan_to_num(g["f_collide"]))
    b = dev(lb, np.zeros_like(f))
    lb.halo(a, 0)
    lb.propagate_collide(a, b, hy, wrap=False)
    out = host(lb, b)
    assert relmax(interior(out, h), interior(fp, h)) < RTOL_F
    # and with index wrap on un-haloed input
    a2 = dev(lb, np.nan_to_num(g["f_collide"]))
    b2 = dev(lb, np.zeros_like(f))
    lb.propagate_collide(a2, b2, hy, wrap=True)
    out2 = host(lb, b2)
    assert np.array_equal(interior(out2, h), interior(out, h))
    lb.free()


# --- bigger seeded cases against the oracle ---------------------------------

CASES = [
    (19, (32, 24, 16), 1, "m10", 0.1, 0.3, (0, 0, 0), False),
    (19, (17, 9, 33), 1, "bgk", 0.07, 0.07, (1e-6, 0, -2e-6), True),
    (19, (16, 16, 16), 2, "trt", 0.1, 0.2, (0, 1e-6, 0), True),
    (27, (20, 12, 28), 1, "m10", 0.1, 0.3, (1e-6, 1e-6, 0), True),
    (27, (8, 8, 8), 1, "bgk", 0.2, 0.2, (0, 0, 0), False),
    (19, (1, 5, 3), 1, "m10", 0.1, 0.3, (0, 0, 0), False),     # degenerate x
    (19, (2, 2, 2), 1, "bgk", 0.1, 0.1, (0, 0, 0), False),
]


@pytest.mark.parametrize("mode", [0, 1, 2, 101, 3], ids=["eager", "fused", "inplace", "fused_soa", "fused_halo"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "q%d-%s-%s" % (c[0], "x".join(map(str, c[1])), c[3]))
def test_seeded_vs_oracle(case, mode):
    import ludwig_amd
    nvel, nlocal, nhalo, scheme, eta, zeta, fbody, ffield = case
    p = lbo.make_param(nvel, nlocal, nhalo, scheme, eta, zeta, 1.0, fbody)
    f0 = lbo.init_synthetic(p)
    nall = lbo.nall(p)
    force = None
    if ffield:
        rng = np.random.default_rng(7)
        force = 1e-6 * rng.standard_normal((3,) + nall)
    nsteps = 6
    # oracle
    f = f0.copy()
    fp = np.zeros_like(f)
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp, force)
    # device
    lb = ludwig_amd.LB(nvel, nlocal, nhalo, mode=mode,
                       halo_scheme=2 if mode else 0)
    lb.relaxation_set(scheme, eta, zeta)
    lb.body_force_set(fbody)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=force)
    lb.lb_memcpy_h2d(f0)
    for _ in range(nsteps):
        lb.step(hy)
    mo = lb.moments()
    out = lb.lb_memcpy_d2h()
    assert relmax(interior(out, nhalo), interior(f, nhalo)) < RTOL_F
    mg = lbo.moments(p, f)
    assert abs(mo[1] - mg[1]) / mg[1] < RTOL_CONSERVED
    gscale = momentum_scale(f, lbo.model(nvel)["cv"], nhalo)
    assert np.max(np.abs(mo[5:8] - mg[5:8])) / gscale < RTOL_CONSERVED
    lb.free()


def test_trt_d3q27_rejected():
    import ludwig_amd
    lb = ludwig_amd.LB(27, (4, 4, 4))
    with pytest.raises(ludwig_amd.LbmiError):
        lb.relaxation_set("trt", 0.1, 0.1)
    lb.free()


# --- generic field halo (hydro_u_halo) ---------------------------------------

@pytest.mark.parametrize("nhalo", [1, 2])
def test_field_halo(nhalo):
    import ludwig_amd
    nlocal = (6, 5, 4)
    lb = ludwig_amd.LB(19, nlocal, nhalo)
    p = lbo.make_param(19, nlocal, nhalo)
    rng = np.random.default_rng(3)
    u = rng.standard_normal((3,) + lbo.nall(p))
    ref = u.copy()
    lbo.halo(p, ref)
    t = dev(lb, u)
    lb.field_halo(t)
    assert np.array_equal(host(lb, t), ref)
    lb.free()


# --- reference regression log on the device ---------------------------------

@pytest.mark.parametrize("mode", [0, 1, 2, 101, 3], ids=["eager", "fused", "inplace", "fused_soa", "fused_halo"])
def test_regression_log_1dp(mode):
    import ludwig_amd
    case = load_expected()["serial-dist-1dp"]
    m = lbo.model(19)
    f0 = initial_f(case, m)
    lb = ludwig_amd.LB(19, tuple(case["size"]), 1, mode=mode)
    lb.relaxation_set(case["scheme"], case["eta"], case["zeta"])
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.lb_memcpy_h2d(f0)
    for _ in range(case["steps"]):
        lb.step(hy)
    mo = lb.moments()
    exp = case["final"]
    mean = mo[1] / mo[0]
    var = abs(mo[2] / mo[0] - mean * mean)
    assert close_as_printed(mo[1], exp["rho_total"], 2)
    assert close_as_printed(mean, exp["rho_mean"], 11)
    assert close_as_printed(mo[3], exp["rho_min"], 11)
    assert close_as_printed(mo[4], exp["rho_max"], 11)
    assert close_as_printed(var, exp["rho_var"], sig=8)
    assert close_as_printed(mo[5], exp["momentum"][0], sig=8)
    u = interior(host(lb, hy.u), 1)
    assert close_as_printed(u[0].min(), exp["u_min"][0], sig=8)
    assert close_as_printed(u[0].max(), exp["u_max"][0], sig=8)
    lb.free()


# --- RCCL ring on one GPU: rank 0 is its own neighbour ----------------------

@pytest.mark.parametrize("packed", [0, 1], ids=["zerocopy", "packed"])
@pytest.mark.parametrize("mode", [0, 1, 3], ids=["eager", "fused", "fused_halo"])
def test_rccl_self_ring(mode, packed):
    """With a 1-rank communicator the X halo goes through pack ->
    ncclSend/ncclRecv (to self) -> unpack instead of the device-side copy;
    the result must be identical."""
    import ludwig_amd
    g = load_golden("q19_m10_fbody")
    meta = g["meta"]
    h = meta["nhalo"]
    res = []
    for use_comm in (False, True):
        lb = make_lb(meta, mode=mode, halo_scheme=2 if mode else 0)
        if use_comm:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
            lb.tune("x_packed", packed)
        hy = make_hydro(lb, g, meta)
        lb.lb_memcpy_h2d(g["f0"])
        for _ in range(meta["nsteps"]):
            lb.step(hy)
        res.append(interior(lb.lb_memcpy_d2h(), h).copy())
        lb.free()
    assert np.array_equal(res[0], res[1])
    assert relmax(res[1], interior(g["f_final"], h)) < RTOL_F


# --- slab decomposition, in one process: N handles, buffers swapped by hand ---

def _slab_setup(nvel, ntotal, nslab, scheme_name, fbody):
    import torch
    import ludwig_amd
    lbs, fa, fb = [], [], []
    for r in range(nslab):
        dec = ludwig_amd.SlabDecomposition(ntotal, nslab, r)
        lb = ludwig_amd.LB(nvel, dec.nlocal, 1, cartsz=nslab, cartrank=r)
        lb.relaxation_set(scheme_name, 0.1, 0.3)
        lb.body_force_set(fbody)
        p = lbo.make_param(nvel, dec.nlocal, 1, scheme_name, 0.1, 0.3, 1.0, fbody)
        f0 = lbo.init_synthetic(p, ntotal, dec.noffset)
        lbs.append(lb)
        fa.append(dev(lb, f0))
        fb.append(torch.zeros_like(fa[-1]))
    torch.cuda.synchronize()
    return lbs, fa, fb


def _slab_exchange(lbs, fa, halo_scheme):
    """pack on every slab, hand the buffers round the periodic ring, unpack:
    what lbmi_x_exchange does with ncclSend/ncclRecv."""
    import torch
    n = len(lbs)
    bufs = []
    for r, lb in enumerate(lbs):
        nlo, nhi = lb.halo_x_count(halo_scheme)
        sendlo = torch.empty(nlo, dtype=torch.float64, device=lb.device)
        sendhi = torch.empty(nhi, dtype=torch.float64, device=lb.device)
        torch.cuda.synchronize()
        lb.halo_x_pack(fa[r], sendlo, sendhi, halo_scheme)
        lb.synchronize()
        bufs.append((sendlo, sendhi))
    for r, lb in enumerate(lbs):
        recvlo = bufs[(r - 1) % n][1]      # prev's last interior plane
        recvhi = bufs[(r + 1) % n][0]      # next's first interior plane
        lb.halo_x_unpack(fa[r], recvlo, recvhi, halo_scheme)
        lb.synchronize()


@pytest.mark.parametrize("fused", [False, True], ids=["eager", "fused"])
@pytest.mark.parametrize("halo_scheme", [0, 2], ids=["full", "reduced"])
@pytest.mark.parametrize("nvel,nslab", [(19, 2), (19, 3), (27, 2)])
def test_slabs_equal_single_domain(nvel, nslab, halo_scheme, fused):
    ntotal = (4 * nslab, 5, 6)
    fbody = (1e-6, 0.0, -1e-6)
    nsteps = 4
    lbs, fa, fb = _slab_setup(nvel, ntotal, nslab, "m10", fbody)

    if not fused:
        for _ in range(nsteps):
            for r, lb in enumerate(lbs):
                lb.collide(fa[r])
                lb.synchronize()
            _slab_exchange(lbs, fa, halo_scheme)
            for r, lb in enumerate(lbs):
                lb.halo_yz(fa[r], halo_scheme)
                lb.propagate(fa[r], fb[r])
                lb.synchronize()
            fa, fb = fb, fa
    else:
        # collide(0); then [exchange X; fused pull+collide with y/z wrap] x
        # (nsteps-1); then exchange, y/z halo, propagate
        for r, lb in enumerate(lbs):
            lb.collide(fa[r])
            lb.synchronize()
        for _ in range(nsteps - 1):
            _slab_exchange(lbs, fa, halo_scheme)
            for r, lb in enumerate(lbs):
                lb.propagate_collide(fa[r], fb[r], None, wrap=True)
                lb.synchronize()
            fa, fb = fb, fa
        _slab_exchange(lbs, fa, halo_scheme)
        for r, lb in enumerate(lbs):
            lb.halo_yz(fa[r], halo_scheme)
            lb.propagate(fa[r], fb[r])
            lb.synchronize()
        fa, fb = fb, fa

    got = np.concatenate([interior(host(lbs[r], fa[r]), 1)
                          for r in range(nslab)], axis=1)
    p = lbo.make_param(nvel, ntotal, 1, "m10", 0.1, 0.3, 1.0, fbody)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp)
    assert relmax(got, interior(f, 1)) < RTOL_F
    for lb in lbs:
        lb.free()


# --- FUSED with the deferred state in the blocked order ----------------------

@pytest.mark.parametrize("nvel,nlocal,nhalo", [(19, (24, 10, 18), 1),
                                                (27, (9, 16, 14), 1),
                                                (19, (12, 12, 12), 2)])
def test_blocked_order_is_invisible(nvel, nlocal, nhalo):
    """FUSED default ("blocked" = 1): the deferred state lives in another order, and
    nothing a caller can observe changes -- a copy out (which flushes) after
    ANY call of ANY step returns what EAGER holds there, bit for bit with
    the unblocked FUSED run, and the run continues undisturbed."""
    import ludwig_amd
    p = lbo.make_param(nvel, nlocal, nhalo, "m10", 0.1, 0.3, 1.0, (1e-6, 0, 2e-6))
    f0 = lbo.init_synthetic(p)
    nsteps = 4

    def run(mode, probe):
        lb = ludwig_amd.LB(nvel, nlocal, nhalo, mode=mode)
        lb.relaxation_set("m10", 0.1, 0.3)
        lb.body_force_set((1e-6, 0, 2e-6))
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        seen, orders = None, []
        for n in range(nsteps):
            for k, call in enumerate((lambda: lb.lb_collide(hy), lb.lb_halo,
                                      lb.lb_propagation)):
                call()
                orders.append(lb.state()[2])
                if probe == (n, k):
                    seen = interior(lb.lb_memcpy_d2h(), nhalo).copy()
                    assert lb.state() == (0, 0, 0)
        final = interior(lb.lb_memcpy_d2h(), nhalo).copy()
        u = host(lb, hy.u)
        lb.free()
        return seen, final, u, orders

    _, ref_final, ref_u, ref_orders = run(101, None)   # FUSED, SoA order
    _, final, u, orders = run(1, None)                 # FUSED, default
    assert 1 not in ref_orders
    assert 1 in orders, "the blocked order was never used"
    assert np.array_equal(final, ref_final)
    assert np.array_equal(u, ref_u)
    for n in range(nsteps):
        for k in range(3):
            ref_seen, ref_final, _, _ = run(0, (n, k))
            seen, final, _, _ = run(1, (n, k))
            assert relmax(seen, ref_seen) < 1e-14, (n, k)
            assert relmax(final, ref_final) < 1e-14, (n, k)


def test_blocked_order_switch_mid_run():
    """Turning the blocked order on and off between steps converts the
    state at the next step or flush."""
    import ludwig_amd
    nlocal = (20, 12, 16)
    p = lbo.make_param(19, nlocal, 1, "bgk", 0.1, 0.1)
    f0 = lbo.init_synthetic(p)
    outs = []
    for toggle in (False, True):
        lb = ludwig_amd.LB(19, nlocal, 1, mode=ludwig_amd.FUSED)
        lb.tune("blocked", 0)
        lb.relaxation_set("bgk", 0.1, 0.1)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        for n in range(8):
            if toggle and n in (2, 5, 6):
                lb.tune("blocked", 1 if n != 5 else 0)
            lb.step(hy)
        outs.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.free()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("concurrent", [0, 1], ids=["serial", "concurrent"])
@pytest.mark.parametrize("nvel,nlocal", [(19, (10, 14, 14)), (27, (6, 14, 30)),
                                         (19, (5, 30, 62))])
def test_rccl_self_ring_blocked_slab(nvel, nlocal, concurrent):
    """Slab path (interior + exchange + boundary planes through a 1-rank RCCL
    ring) with the deferred state in the blocked order: lattices whose tail
    past the last whole block lies in the never-pulled end of the high halo
    plane. Identical to EAGER on the same ring, flush at any point."""
    import ludwig_amd
    p = lbo.make_param(nvel, nlocal, 1, "m10", 0.1, 0.3, 1.0, (1e-6, 0, 0))
    f0 = lbo.init_synthetic(p)
    res, orders = [], []
    for mode in (ludwig_amd.EAGER, ludwig_amd.FUSED, ludwig_amd.FUSED_SOA):
        lb = ludwig_amd.LB(nvel, nlocal, 1, mode=mode, halo_scheme=2)
        lb.relaxation_set("m10", 0.1, 0.3)
        lb.body_force_set((1e-6, 0, 0))
        lb.comm_init(ludwig_amd.LB.comm_unique_id())
        lb.tune("x_concurrent", concurrent)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        seen = []
        for n in range(5):
            lb.step(hy)
            seen.append(lb.state()[2])
            if n == 2:
                mid = interior(lb.lb_memcpy_d2h(), 1).copy()     # flushes
        res.append((mid, interior(lb.lb_memcpy_d2h(), 1).copy(),
                    host(lb, hy.u)))
        orders.append(seen)
        lb.free()
    assert 1 in orders[1] and 1 not in orders[2]
    for k in (1, 2):
        assert relmax(res[k][0], res[0][0]) < 1e-14
        assert relmax(res[k][1], res[0][1]) < 1e-14
    assert np.array_equal(res[1][1], res[2][1])
    assert np.array_equal(res[1][2], res[2][2])
    # and against the oracle
    f = f0.copy()
    fp = np.zeros_like(f)
    for _ in range(5):
        f, fp = lbo.step(p, f, fp)
    assert relmax(res[1][1], interior(f, 1)) < RTOL_F


# --- viscosity model: local relaxation times from hydro->eta ------------------

from tests.common import golden_visc_names  # noqa: E402


@pytest.mark.parametrize("mode", [0, 1, 2, 101, 3], ids=["eager", "fused", "inplace", "fused_soa", "fused_halo"])
@pytest.mark.parametrize("name", golden_visc_names())
def test_local_viscosity_vs_reference(name, mode):
    """lb_collide with visc != NULL (collision.c:386-404): the rates of every
    site from hydro->eta, bulk viscosity in the Newtonian ratio."""
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = make_lb(meta, mode=mode)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=g["force"], eta=g["eta"])
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_collide(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(g["f_collide"], h)) < RTOL_F
    assert relmax(interior(host(lb, hy.u), h), interior(g["u"], h)) < RTOL_F
    lb.lb_memcpy_h2d(g["f0"])
    for _ in range(meta["nsteps"]):
        lb.step(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(g["f_final"], h)) < RTOL_F
    lb.free()


def test_local_viscosity_seeded_blocked():
    """A lattice of whole blocks: the fused blocked kernel with hydro->eta,
    against the oracle."""
    import ludwig_amd
    nlocal = (10, 14, 14)
    p = lbo.make_param(19, nlocal, 1, "trt", 0.1, 0.25)
    f0 = lbo.init_synthetic(p)
    rng = np.random.default_rng(4)
    eta = 0.1 * (1.0 + 0.4 * rng.random(lbo.nall(p)))
    f = f0.copy()
    fp = np.zeros_like(f)
    for _ in range(4):
        lbo.collide_visc(p, f, None, None, eta)
        lbo.halo(p, f)
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    lb = ludwig_amd.LB(19, nlocal, 1, mode=ludwig_amd.FUSED)
    lb.relaxation_set("trt", 0.1, 0.25)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, eta=eta)
    lb.lb_memcpy_h2d(f0)
    lb.run(hy, 4)
    assert lb.state()[2] == 1
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f, 1)) < RTOL_F
    lb.free()
