"""Row f2 on the device: width-2 field halo, 7- and 27-point gradients, the
symmetric free-energy force and the Cahn-Hilliard step (advection orders
1..4), against the compiled-reference fixtures and the
oracle. Needs an MI355X."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                        # noqa: E402
from tests.common import golden_fe_names, interior, load_golden, relmax  # noqa: E402


def _dev(lb, a):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).to(lb.device)
    torch.cuda.synchronize(lb.device)
    return t


def _host(lb, t):
    lb.synchronize()
    return t.cpu().numpy()


def _scheme(lb, meta):
    """The fixture's fd_gradient_calculation / fd_advection_scheme_order."""
    lb.fe_scheme_set(meta.get("grad_npt", 7), meta.get("advection_order", 1))


@pytest.mark.parametrize("name", golden_fe_names())
def test_field_halo_width2_exact(name):
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), h)
    phi = np.zeros_like(g["phi"])
    interior(phi, h)[...] = interior(g["phi"], h)
    t = _dev(lb, phi)
    lb.field_halo_n(t, 2)
    assert np.array_equal(_host(lb, t), g["phi"])
    lb.free()


@pytest.mark.parametrize("name", golden_fe_names())
def test_gradient_exact(name):
    import ludwig_amd
    import torch
    g = load_golden(name)
    meta = g["meta"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), meta["nhalo"])
    _scheme(lb, meta)
    phi = _dev(lb, g["phi"])
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize()
    lb.field_grad(phi, grad, delsq)
    s = (slice(1, -1),) * 3
    # 7-point: differences of two doubles times 0.5, no contraction possible,
    # so bit-exact; the Laplacians end in "- n*phi" which may contract
    if meta.get("grad_npt", 7) == 7:
        assert np.array_equal(_host(lb, grad)[(slice(None),) + s],
                              g["grad"][(slice(None),) + s])
    assert relmax(_host(lb, grad)[(slice(None),) + s],
                  g["grad"][(slice(None),) + s]) < 1e-15
    assert relmax(_host(lb, delsq)[s], g["delsq"][s]) < 1e-15
    lb.free()


@pytest.mark.parametrize("from_grad", [True, False], ids=["from_grad", "from_phi"])
@pytest.mark.parametrize("name", golden_fe_names())
def test_symmetric_force_vs_reference(name, from_grad):
    import ludwig_amd
    import torch
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), h)
    _scheme(lb, meta)
    phi = _dev(lb, g["phi"])
    force = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize()
    if from_grad:
        lb.symmetric_force(meta["a"], meta["b"], meta["kappa"], phi, force,
                           _dev(lb, g["grad"]), _dev(lb, g["delsq"]))
    else:
        lb.symmetric_force(meta["a"], meta["b"], meta["kappa"], phi, force)
    f = _host(lb, force)
    assert relmax(interior(f, h), interior(g["force"], h)) < 1e-12
    # it ADDS to the force field: a second call doubles it
    if from_grad:
        lb.symmetric_force(meta["a"], meta["b"], meta["kappa"], phi, force,
                           _dev(lb, g["grad"]), _dev(lb, g["delsq"]))
        assert relmax(interior(_host(lb, force), h), 2 * interior(g["force"], h)) < 1e-12
    lb.free()


def test_force_chain_seeded_vs_oracle_and_feeds_collision():
    """phi -> halo(2) -> force (from phi) -> lb_collide reads it: the
    coupling of BASELINE config 4, against the oracle end to end."""
    import ludwig_amd
    import torch
    nlocal, h = (24, 10, 18), 2
    a, b, kappa = -0.00625, 0.00625, 0.004
    p = lbo.make_param(19, nlocal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(5)
    phi = np.zeros(lbo.nall(p))
    interior(phi, h)[...] = 0.5 * rng.standard_normal(nlocal)
    # oracle
    phi_o = phi.copy()
    lbo.field_halo(p, phi_o, 2)
    grad, delsq = lbo.grad_7pt(p, phi_o)
    force_o = np.zeros((3,) + phi.shape)
    lbo.symm_force(p, a, b, kappa, phi_o, grad, delsq, force_o)
    f0 = lbo.init_synthetic(p)
    f_o = f0.copy()
    lbo.collide(p, f_o, force_o)
    # device
    lb = ludwig_amd.LB(19, nlocal, h)
    lb.relaxation_set("m10", 0.1, 0.3)
    t = _dev(lb, phi)
    lb.field_halo_n(t, 2)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + phi.shape))
    lb.symmetric_force(a, b, kappa, t, hy.force)
    assert relmax(interior(_host(lb, hy.force), h), interior(force_o, h)) < 1e-12
    lb.lb_memcpy_h2d(f0)
    lb.lb_collide(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(f_o, h)) < 1e-12
    lb.free()


@pytest.mark.parametrize("from_delsq", [True, False], ids=["from_delsq", "from_phi"])
@pytest.mark.parametrize("name", golden_fe_names())
def test_cahn_hilliard_vs_reference(name, from_delsq):
    import ludwig_amd
    import torch
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), h)
    _scheme(lb, meta)
    phi = _dev(lb, g["phi"])
    # u: interior from the fixture, halo by our own width-1 field halo
    u0 = np.zeros_like(g["u"])
    interior(u0, h)[...] = interior(g["u"], h)
    u = _dev(lb, u0)
    lb.field_halo_n(u, 1)
    out = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize()
    lb.cahn_hilliard(meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                     phi, u, out, _dev(lb, g["delsq"]) if from_delsq else None)
    res = _host(lb, out)
    assert relmax(interior(res, h), interior(g["phi_new"], h)) < 1e-12
    assert abs(interior(res, h).sum() - interior(g["phi"], h).sum()) < 1e-12
    lb.free()


def test_cahn_hilliard_rejects_aliasing():
    import ludwig_amd
    import torch
    lb = ludwig_amd.LB(19, (4, 4, 4), 2)
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    u = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.cahn_hilliard(-1.0, 1.0, 1.0, 0.1, phi, u, phi)
    lb.free()


@pytest.mark.parametrize("npt,order", [(7, 1), (27, 2), (27, 3), (7, 4)])
def test_binary_fluid_steps_vs_oracle(npt, order):
    """A few complete steps of BASELINE config 4 on a small box, the order
    of ludwig.c:537-860: f_zero, phi halo, force, Cahn-Hilliard (with the u
    of the previous collision), u_zero, collide, halo, propagate -- device
    (fused-from-phi kernels, FUSED LB step) against the oracle."""
    import ludwig_amd
    import torch
    nlocal, h, nsteps = (12, 10, 8), 2, 4
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(19, nlocal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(9)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(nlocal)
    f0 = lbo.init_synthetic(p)

    # oracle
    phi = phi0.copy()
    f = f0.copy()
    fp = np.zeros_like(f)
    u = np.zeros((3,) + phi.shape)
    rho = np.zeros(phi.shape)
    for _ in range(nsteps):
        force = np.zeros((3,) + phi.shape)
        lbo.field_halo(p, phi, 2)
        grad, delsq = lbo.grad(p, phi, npt)
        lbo.symm_force(p, a, b, kappa, phi, grad, delsq, force)
        lbo.field_halo(p, u, 1)
        lbo.cahn_hilliard(p, a, b, kappa, mob, phi, delsq, u, order=order)
        u[...] = 0.0
        f, fp = lbo.step(p, f, fp, force, None, rho, u)

    # device
    lb = ludwig_amd.LB(19, nlocal, h, mode=ludwig_amd.FUSED)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.fe_scheme_set(npt, order)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + phi.shape))
    pa = _dev(lb, phi0)
    pb = torch.zeros_like(pa)
    lb.lb_memcpy_h2d(f0)
    for _ in range(nsteps):
        lb.hydro_field_set(hy.force, (0, 0, 0))
        lb.field_halo_n(pa, 2)
        lb.symmetric_force(a, b, kappa, pa, hy.force)
        lb.field_halo_n(hy.u, 1)
        lb.cahn_hilliard(a, b, kappa, mob, pa, hy.u, pb)
        pa, pb = pb, pa
        lb.hydro_field_set(hy.u, (0, 0, 0))
        lb.step(hy)
    assert relmax(interior(_host(lb, pa), h), interior(phi, h)) < 1e-12
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(f, h)) < 1e-12
    lb.free()


@pytest.mark.parametrize("lazy", [0, 2])
def test_one_kernel_binary_fluid_step_with_rho_on_demand(lazy):
    """hydro_lazy 2 (what the binding sets where a free energy needs u every
    step): the one-kernel step stores u and leaves rho until it is asked for;
    asked after any step (lbmi_lb_hydro_sync) it is the oracle's, and the run
    is the same bit for bit as with rho stored every step."""
    import ludwig_amd
    import torch
    nlocal, h, nsteps = (12, 10, 8), 2, 5
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(19, nlocal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(12)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(nlocal)
    f0 = lbo.init_synthetic(p)
    lb = ludwig_amd.LB(19, nlocal, h, mode=ludwig_amd.FUSED)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.fe_scheme_set(7, 1)
    lb.tune("hydro_lazy", lazy)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    ua, ub = hy.u, torch.zeros_like(hy.u)
    pa, pb = _dev(lb, phi0), torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    lb.lb_memcpy_h2d(f0)
    phi = phi0.copy()
    f = f0.copy()
    fp = np.zeros_like(f)
    u = np.zeros((3,) + phi.shape)
    rho = np.zeros(phi.shape)
    for n in range(nsteps):
        force = np.zeros((3,) + phi.shape)
        lbo.field_halo(p, phi, 2)
        grad, delsq = lbo.grad(p, phi, 7)
        lbo.symm_force(p, a, b, kappa, phi, grad, delsq, force)
        lbo.field_halo(p, u, 1)
        lbo.cahn_hilliard(p, a, b, kappa, mob, phi, delsq, u, order=1)
        u[...] = 0.0
        f, fp = lbo.step(p, f, fp, force, None, rho, u)
        hy.u = ub if n % 2 == 0 else ua
        lb.symmetric_lb_step(hy, ua if n % 2 == 0 else ub, a, b, kappa, mob, pa, pb)
        pa, pb = pb, pa
        if n in (1, nsteps - 1):
            lb.hydro_sync()
            lb.synchronize()
            torch.cuda.synchronize()
            assert relmax(interior(hy.rho.cpu().numpy(), h), interior(rho, h)) < 1e-12, n
        assert relmax(interior(_host(lb, hy.u), h), interior(u, h)) < 1e-12, n
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(f, h)) < 1e-12
    assert relmax(interior(_host(lb, pa), h), interior(phi, h)) < 1e-12
    lb.free()


@pytest.mark.parametrize("nvel,scheme,order", [(19, "m10", 1), (19, "m10", 2), (19, "m10", 3),
                                               (19, "m10", 4), (19, "bgk", 1), (19, "bgk", 3),
                                               (27, "m10", 1), (27, "bgk", 2), (27, "m10", 4)])
@pytest.mark.parametrize("split", [False, True], ids=["step", "collide"])
def test_one_kernel_binary_fluid_step_vs_oracle(nvel, order, scheme, split):
    """(split: lbmi_symmetric_lb_collide, then lb_halo and lb_propagation by
    the caller -- the form the binding uses, ludwig.c calls those two itself.)
    lbmi_symmetric_lb_step: the whole step of BASELINE config 4 as ONE
    kernel (force and Cahn-Hilliard update of a site evaluated by the thread
    that collides it, u of the previous step from a second array) against
    the oracle running the reference's order of calls (ludwig.c:537-860):
    phi, the distributions, rho and u after every step, advection orders 1-4.
    The first step after the copy-in runs the separate calls (nothing is
    pending yet), the others the fused kernel: lbmi_lb_state says which."""
    import ludwig_amd
    import torch
    nlocal, h, nsteps = (12, 10, 8), 2, 5
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(nvel, nlocal, h, scheme, 0.1, 0.3, 1.0, (1e-6, -2e-6, 5e-7))
    rng = np.random.default_rng(9)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(nlocal)
    f0 = lbo.init_synthetic(p)

    lb = ludwig_amd.LB(nvel, nlocal, h, mode=ludwig_amd.FUSED)
    lb.relaxation_set(scheme, 0.1, 0.3)
    lb.body_force_set((1e-6, -2e-6, 5e-7))
    lb.fe_scheme_set(7, order)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    ua = hy.u
    ub = torch.zeros_like(ua)
    pa = _dev(lb, phi0)
    pb = torch.zeros_like(pa)
    lb.lb_memcpy_h2d(f0)

    phi = phi0.copy()
    f = f0.copy()
    fp = np.zeros_like(f)
    u = np.zeros((3,) + phi.shape)
    rho = np.zeros(phi.shape)
    for n in range(nsteps):
        # oracle: the reference's sequence
        force = np.zeros((3,) + phi.shape)
        lbo.field_halo(p, phi, 2)
        grad, delsq = lbo.grad(p, phi, 7)
        lbo.symm_force(p, a, b, kappa, phi, grad, delsq, force)
        lbo.field_halo(p, u, 1)
        lbo.cahn_hilliard(p, a, b, kappa, mob, phi, delsq, u, order=order)
        u[...] = 0.0
        f, fp = lbo.step(p, f, fp, force, None, rho, u)
        # device: one call; u_prev = what the last call stored
        pending = lb.state()[1]
        hy.u = ub if n % 2 == 0 else ua
        if split:
            lb.symmetric_lb_collide(hy, ua if n % 2 == 0 else ub, a, b, kappa, mob, pa, pb)
            assert lb.state()[:2] == (0, 0)
            lb.lb_halo()
            lb.lb_propagation()
        else:
            lb.symmetric_lb_step(hy, ua if n % 2 == 0 else ub, a, b, kappa, mob, pa, pb)
        pa, pb = pb, pa
        assert pending == (1 if n > 0 else 0)
        lb.synchronize()
        torch.cuda.synchronize()
        assert relmax(interior(_host(lb, pa), h), interior(phi, h)) < 1e-12, n
        assert relmax(interior(hy.u.cpu().numpy(), h), interior(u, h)) < 1e-12, n
        assert relmax(interior(hy.rho.cpu().numpy(), h), interior(rho, h)) < 1e-12, n
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(f, h)) < 1e-12
    lb.free()


@pytest.mark.parametrize("mode", ["eager", "fused_halo", "fused_soa", "inplace"])
def test_binary_fluid_step_off_the_fused_route_is_the_same(mode):
    """lbmi_symmetric_lb_step where the one-kernel form does not apply (other
    execution modes; 27-point gradients; the SoA order of the deferred state is
    a variant of the fused kernel): the separate calls behind the same entry
    point, against the fused route of a FUSED handle."""
    import ludwig_amd
    import torch
    nlocal, h, nsteps = (8, 12, 9), 2, 4
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(19, nlocal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(10)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(nlocal)
    f0 = lbo.init_synthetic(p)
    modes = {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO,
             "fused_soa": ludwig_amd.FUSED_SOA, "inplace": ludwig_amd.INPLACE}
    out = []
    for m in (ludwig_amd.FUSED, modes[mode]):
        lb = ludwig_amd.LB(19, nlocal, h, mode=m)
        lb.relaxation_set("m10", 0.1, 0.3)
        lb.fe_scheme_set(7, 2)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        ua, ub = hy.u, torch.zeros_like(hy.u)
        pa, pb = _dev(lb, phi0), torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        lb.lb_memcpy_h2d(f0)
        for n in range(nsteps):
            hy.u = ub if n % 2 == 0 else ua
            lb.symmetric_lb_step(hy, ua if n % 2 == 0 else ub, a, b, kappa, mob, pa, pb)
            pa, pb = pb, pa
        lb.synchronize()
        torch.cuda.synchronize()
        out.append((interior(_host(lb, pa), h).copy(), interior(hy.u.cpu().numpy(), h).copy(),
                    interior(lb.lb_memcpy_d2h(), h).copy()))
        lb.free()
    for x, y in zip(out[0], out[1]):
        assert relmax(x, y) < 1e-13


def test_binary_fluid_step_arguments():
    import ludwig_amd
    import torch
    lb = ludwig_amd.LB(19, (8, 8, 8), 2, mode=ludwig_amd.FUSED)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.ones((3,) + lb.nall))
    pa = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    pb = torch.zeros_like(pa)
    ub = torch.zeros_like(hy.u)
    with pytest.raises(ludwig_amd.LbmiError):          # u aliases u_prev
        lb.symmetric_lb_step(hy, hy.u, -1.0, 1.0, 1.0, 0.1, pa, pb)
    with pytest.raises(ludwig_amd.LbmiError):          # phi_out aliases phi
        lb.symmetric_lb_step(hy, ub, -1.0, 1.0, 1.0, 0.1, pa, pa)
    with pytest.raises(ludwig_amd.LbmiError):          # a force field with contents
        lb.symmetric_lb_step(hy, ub, -1.0, 1.0, 1.0, 0.1, pa, pb)
    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))      # known zero: accepted
    lb.symmetric_lb_step(hy, ub, -1.0, 1.0, 1.0, 0.1, pa, pb)
    lb.free()


@pytest.mark.parametrize("name", golden_fe_names())
def test_symmetric_step_equals_separate_kernels(name):
    import ludwig_amd
    import torch
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), h)
    _scheme(lb, meta)
    phi = _dev(lb, g["phi"])
    u0 = np.zeros_like(g["u"])
    interior(u0, h)[...] = interior(g["u"], h)
    u = _dev(lb, u0)
    lb.field_halo_n(u, 1)
    args = (meta["a"], meta["b"], meta["kappa"])
    f1 = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    f2 = torch.zeros_like(f1)
    o1 = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    o2 = torch.zeros_like(o1)
    torch.cuda.synchronize()
    lb.symmetric_force(*args, phi, f1)
    lb.cahn_hilliard(*args, meta["mobility"], phi, u, o1)
    lb.symmetric_step(*args, meta["mobility"], phi, u, f2, o2)
    assert relmax(interior(_host(lb, f2), h), interior(g["force"], h)) < 1e-12
    assert relmax(interior(_host(lb, o2), h), interior(g["phi_new"], h)) < 1e-12
    assert relmax(_host(lb, f2), _host(lb, f1)) < 1e-14
    assert relmax(_host(lb, o2), _host(lb, o1)) < 1e-14
    # overwrite mode: same force regardless of what was there
    f3 = torch.full_like(f1, 123.0)
    torch.cuda.synchronize()
    lb.symmetric_step(*args, meta["mobility"], phi, u, f3, o2, accumulate=False)
    assert np.array_equal(interior(_host(lb, f3), h), interior(_host(lb, f2), h))
    # the same pass fed from the gradient arrays of field_grad
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    f4 = torch.zeros_like(f1)
    o4 = torch.zeros_like(o1)
    torch.cuda.synchronize()
    lb.field_grad(phi, grad, delsq)
    lb.symmetric_step_grad(*args, meta["mobility"], phi, grad, delsq, u, f4, o4)
    assert relmax(interior(_host(lb, f4), h), interior(g["force"], h)) < 1e-12
    assert relmax(interior(_host(lb, o4), h), interior(g["phi_new"], h)) < 1e-12
    lb.free()


def test_fe_scheme_rejects_what_the_reference_rejects():
    import ludwig_amd
    lb = ludwig_amd.LB(19, (4, 4, 4), 2)
    for npt, order in ((9, 1), (7, 0), (7, 5), (27, 7)):
        with pytest.raises(ludwig_amd.LbmiError):
            lb.fe_scheme_set(npt, order)
    lb.free()
    lb = ludwig_amd.LB(19, (4, 4, 4), 1)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.fe_scheme_set(7, 3)          # reaches two sites: needs nhalo 2
    lb.fe_scheme_set(27, 2)
    lb.free()


@pytest.mark.parametrize("route", ["from_phi", "from_grad", "separate"])
@pytest.mark.parametrize("name", ["iodrop-mpi1-io1", "serial-symm-dr1"])
def test_drop_regression_log(name, route):
    """BASELINE config 4 end to end against the reference's own regression
    logs (a relaxing droplet: 27-point gradients, second-order advection,
    stress-divergence force, Cahn-Hilliard, M10 collision with the force,
    20 coupled steps): every printed statistic to its printed precision."""
    import ludwig_amd
    import torch
    from tests.regression_cases import (check_drop_report, drop_phi,
                                        drop_report, load_expected_drop, rest_f)
    case = load_expected_drop()[name]
    report_at = sorted(int(k) for k in case["reports"])
    h = 2
    a, b, kappa, mob = case["a"], case["b"], case["kappa"], case["mobility"]
    lb = ludwig_amd.LB(19, tuple(case["size"]), h, mode=ludwig_amd.FUSED)
    lb.relaxation_set("m10", case["eta"], case["zeta"])
    lb.fe_scheme_set(case["grad_npt"], case["advection_order"])
    phi0 = drop_phi(case, h)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + phi0.shape))
    pa = _dev(lb, phi0)
    pb = torch.zeros_like(pa)
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    lb.lb_memcpy_h2d(rest_f(ludwig_amd.model(19), lb.nall, h))
    torch.cuda.synchronize()
    for step in range(1, max(report_at) + 1):
        lb.field_halo_n(pa, 2)
        lb.field_halo_n(hy.u, 1)
        if route == "from_phi":
            lb.symmetric_step(a, b, kappa, mob, pa, hy.u, hy.force, pb,
                              accumulate=False)
        elif route == "from_grad":
            lb.field_grad(pa, grad, delsq)
            lb.symmetric_step_grad(a, b, kappa, mob, pa, grad, delsq, hy.u,
                                   hy.force, pb, accumulate=False)
        else:
            lb.hydro_field_set(hy.force, (0, 0, 0))
            lb.field_grad(pa, grad, delsq)
            lb.symmetric_force(a, b, kappa, pa, hy.force, grad, delsq)
            lb.cahn_hilliard(a, b, kappa, mob, pa, hy.u, pb, delsq)
        if step in report_at and route == "from_phi":
            lb.field_grad(pa, grad, delsq)        # of the start-of-step phi
        pa, pb = pb, pa
        lb.hydro_field_set(hy.u, (0, 0, 0))
        lb.step(hy)
        if step in report_at:
            rep = drop_report(case, _host(lb, pa), _host(lb, grad),
                              lb.moments(), _host(lb, hy.u), h)
            check_drop_report(rep, case["reports"][str(step)])
    lb.free()


@pytest.mark.parametrize("nel", [1, 3])
def test_field_halo_width2_over_rccl_ring(nel):
    """Width-2 field halo with the X planes of both layers through a 1-rank
    RCCL ring = the device-side periodic copy, bit for bit."""
    import ludwig_amd
    import torch
    nlocal = (7, 6, 5)
    rng = np.random.default_rng(8)
    out = []
    for ring in (False, True):
        lb = ludwig_amd.LB(19, nlocal, 2)
        if ring:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
        a = np.zeros((nel,) + lb.nall)
        rng2 = np.random.default_rng(8)
        interior(a, 2)[...] = rng2.random((nel,) + nlocal)
        t = _dev(lb, a if nel > 1 else a[0])
        torch.cuda.synchronize()
        lb.field_halo_n(t, 2)
        out.append(_host(lb, t).copy())
        lb.free()
    assert np.array_equal(out[0], out[1])
    assert np.count_nonzero(out[0]) == out[0].size


def test_binary_fluid_steps_through_rccl_ring():
    """The whole config-4 step with every X halo (phi width 2, u width 1, the
    distributions) over a 1-rank RCCL ring: the slab form of the binary
    fluid, against the oracle."""
    import ludwig_amd
    import torch
    nlocal, h, nsteps = (10, 12, 12), 2, 4
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(19, nlocal, h, "m10", 0.1, 0.3)
    rng = np.random.default_rng(19)
    phi0 = np.zeros(lbo.nall(p))
    interior(phi0, h)[...] = 0.1 * rng.standard_normal(nlocal)
    f0 = lbo.init_synthetic(p)
    phi = phi0.copy()
    f = f0.copy()
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

    lb = ludwig_amd.LB(19, nlocal, h, mode=ludwig_amd.FUSED, halo_scheme=2)
    lb.comm_init(ludwig_amd.LB.comm_unique_id())
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.fe_scheme_set(7, 2)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + phi.shape))
    pa = _dev(lb, phi0)
    pb = torch.zeros_like(pa)
    lb.lb_memcpy_h2d(f0)
    for _ in range(nsteps):
        lb.field_halo_n(pa, 2)
        lb.field_halo_n(hy.u, 1)
        lb.symmetric_step(a, b, kappa, mob, pa, hy.u, hy.force, pb, accumulate=False)
        pa, pb = pb, pa
        lb.step(hy)
    assert relmax(interior(_host(lb, pa), h), interior(phi, h)) < 1e-12
    assert relmax(interior(lb.lb_memcpy_d2h(), h), interior(f, h)) < 1e-12
    lb.free()


@pytest.mark.parametrize("npt,order", [(7, 1), (27, 2), (7, 3), (27, 4)])
@pytest.mark.parametrize("nlocal", [(12, 10, 8), (33, 9, 34), (4, 3, 2)])
def test_symmetric_step_periodic_equals_halo_swapped(nlocal, npt, order):
    """The pass that wraps the periodic box by index = field halos of phi
    (2 layers) and u (1 layer) followed by the plain pass, bit for bit."""
    import ludwig_amd
    import torch
    h = 2
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    lb = ludwig_amd.LB(19, nlocal, h)
    lb.fe_scheme_set(npt, order)
    rng = np.random.default_rng(31)
    phi0 = np.zeros(lb.nall)
    interior(phi0, h)[...] = 0.3 * rng.standard_normal(nlocal)
    u0 = np.zeros((3,) + lb.nall)
    interior(u0, h)[...] = 0.02 * rng.standard_normal((3,) + nlocal)
    # halo-swapped form
    p1, uu1 = _dev(lb, phi0), _dev(lb, u0)
    f1 = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    o1 = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize()
    lb.field_halo_n(p1, 2)
    lb.field_halo_n(uu1, 1)
    lb.symmetric_step(a, b, kappa, mob, p1, uu1, f1, o1, accumulate=False)
    # wrap form: halos left as they are (zeros, and then garbage)
    p2, uu2 = _dev(lb, phi0), _dev(lb, u0)
    p2[:h] = 77.0
    uu2[:, :, :h] = -55.0
    f2 = torch.zeros_like(f1)
    o2 = torch.zeros_like(o1)
    torch.cuda.synchronize()
    lb.symmetric_step_periodic(a, b, kappa, mob, p2, uu2, f2, o2, accumulate=False)
    assert np.array_equal(interior(_host(lb, f2), h), interior(_host(lb, f1), h))
    assert np.array_equal(interior(_host(lb, o2), h), interior(_host(lb, o1), h))
    lb.free()
