"""Row f4 on the device: the two-distribution (symmetric_lb) step --
phi_lb_to_field, lb_collision_binary, lb_halo and lb_propagation of both
distributions -- through the C-ABI against the compiled-reference fixtures
and the oracle. Needs an MI355X."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                                    # noqa: E402
from tests.common import golden_binary_names, interior, load_golden, relmax  # noqa: E402


def _lb(meta, scheme="m10", mode=0):
    import ludwig_amd
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), 1, ndist=2, mode=mode)
    lb.relaxation_set(scheme, meta["eta"], meta["zeta"])
    lb.body_force_set(meta["fbody"])
    return lb


def _dev(lb, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(lb.device)


def _host(lb, t):
    lb.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("name", golden_binary_names())
def test_phi_to_field_exact(name):
    import torch
    g = load_golden(name)
    lb = _lb(g["meta"])
    lb.lb_memcpy_h2d(g["f0"])
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    lb.phi_to_field(phi)
    assert np.array_equal(interior(_host(lb, phi), 1), interior(g["phi"], 1))
    lb.free()


@pytest.mark.parametrize("name", golden_binary_names())
def test_binary_collision_vs_reference(name):
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    lb = _lb(meta)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                         _dev(lb, g["phi"]), _dev(lb, g["grad"]), _dev(lb, g["delsq"]))
    f = lb.lb_memcpy_d2h()
    nv = meta["nvel"]
    assert relmax(interior(f[:nv], 1), interior(g["f_collide"][:nv], 1)) < 1e-12
    assert relmax(interior(f[nv:], 1), interior(g["f_collide"][nv:], 1)) < 1e-12
    assert relmax(interior(_host(lb, hy.u), 1), interior(g["u"], 1)) < 1e-12
    lb.free()


@pytest.mark.parametrize("mode", [0, 3, 1], ids=["eager", "fused_halo", "fused"])
@pytest.mark.parametrize("name", golden_binary_names())
def test_binary_steps_vs_reference(name, mode):
    """Whole steps as ludwig.c runs them with free_energy symmetric_lb.
    FUSED_HALO: the propagation of both distributions is deferred into the
    next collision, and phi_lb_to_field pulls from the pending state. FUSED:
    the halo swap of both is deferred as well: the pulls of the collision and
    of phi_lb_to_field wrap the periodic box by index."""
    import ludwig_amd
    import torch
    g = load_golden(name)
    meta = g["meta"]
    lb = _lb(meta, mode=mode)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.fe_scheme_set(27, 1)
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    lb.lb_memcpy_h2d(g["f0"])
    for _ in range(meta["nsteps"]):
        lb.phi_to_field(phi)
        lb.field_halo_n(phi, 1)
        lb.field_grad(phi, grad, delsq)
        lb.hydro_field_set(hy.u, (0, 0, 0))
        lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"],
                             meta["mobility"], phi, grad, delsq)
        lb.lb_halo()
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, 1), interior(g["f_final"], 1)) < 1e-12
    # moments look at the density distribution
    mo = lb.moments()
    nv = meta["nvel"]
    assert abs(mo[1] - interior(g["f_final"][:nv], 1).sum()) < 1e-10
    lb.free()


@pytest.mark.parametrize("mode", [0, 3, 1], ids=["eager", "fused_halo", "fused"])
@pytest.mark.parametrize("nvel,scheme", [(19, "bgk"), (19, "trt"), (27, "bgk")])
def test_binary_seeded_vs_oracle(nvel, scheme, mode):
    """Larger box, other relaxation schemes, a force field: oracle parity."""
    import ludwig_amd
    import torch
    nlocal = (20, 12, 16)
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(nvel, nlocal, 1, scheme, 0.1, 0.2, 1.0, (1e-6, 0, 0))
    rng = np.random.default_rng(21)
    f2 = np.zeros((2 * nvel,) + lbo.nall(p))
    f2[:nvel] = lbo.init_synthetic(p)
    w = lbo.model(nvel)["wv"]
    ph0 = 0.3 * rng.standard_normal(nlocal)
    for q in range(nvel):
        interior(f2[nvel + q], 1)[...] = w[q] * ph0 * (1 + 0.01 * rng.standard_normal(nlocal))
    force = 1e-6 * rng.standard_normal((3,) + lbo.nall(p))
    f0 = f2.copy()
    fp2 = np.zeros_like(f2)
    u = np.zeros((3,) + lbo.nall(p))
    for _ in range(3):
        f2, fp2, _, _, _ = lbo.step_binary(p, f2, fp2, a, b, kappa, mob, force, u)

    lb = ludwig_amd.LB(nvel, nlocal, 1, ndist=2, mode=mode)
    lb.relaxation_set(scheme, 0.1, 0.2)
    lb.body_force_set((1e-6, 0, 0))
    lb.fe_scheme_set(27, 1)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=force)
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    lb.lb_memcpy_h2d(f0)
    for _ in range(3):
        lb.phi_to_field(phi)
        lb.field_halo_n(phi, 1)
        lb.field_grad(phi, grad, delsq)
        lb.lb_collide_binary(hy, a, b, kappa, mob, phi, grad, delsq)
        lb.lb_halo()
        lb.lb_propagation()
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f2, 1)) < 1e-12
    assert relmax(interior(_host(lb, hy.u), 1), interior(u, 1)) < 1e-12
    lb.free()


def test_binary_rejections():
    import ludwig_amd
    with pytest.raises(ludwig_amd.LbmiError):
        ludwig_amd.LB(19, (4, 4, 4), 1, ndist=2, mode=ludwig_amd.INPLACE)
    with pytest.raises(ludwig_amd.LbmiError):
        ludwig_amd.LB(19, (4, 4, 4), 1, ndist=3)
    lb = ludwig_amd.LB(19, (4, 4, 4), 1, ndist=2)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.lb_collide(hy)                      # the single-fluid entry point
    lb.free()
    lb = ludwig_amd.LB(19, (4, 4, 4), 1)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.phi_to_field(hy.rho)                # needs ndist = 2
    lb.free()


# --- one distribution, fe->use_stress_relaxation -----------------------------

from tests.common import golden_relax_names  # noqa: E402


@pytest.mark.parametrize("mode", [0, 1, 3], ids=["eager", "fused", "fused_halo"])
@pytest.mark.parametrize("name", golden_relax_names())
def test_stress_relaxation_vs_reference(name, mode):
    """fe->use_stress_relaxation (collision.c:413-429). fused / fused_halo:
    from the second step on the propagation of the step before runs inside
    the collision (k_propagate_collide_fe: one pass over f; it used to flush
    every step) -- lbmi_lb_state says so."""
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), 1, mode=mode)
    lb.relaxation_set("m10", meta["eta"], meta["zeta"])
    lb.body_force_set(meta["fbody"])
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    phi, grad, delsq = _dev(lb, g["phi"]), _dev(lb, g["grad"]), _dev(lb, g["delsq"])
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_collide_fe(hy, meta["a"], meta["b"], meta["kappa"], phi, grad, delsq)
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(g["f_collide"], 1)) < 1e-12
    assert relmax(interior(_host(lb, hy.rho), 1), interior(g["rho"], 1)) < 1e-12
    assert relmax(interior(_host(lb, hy.u), 1), interior(g["u"], 1)) < 1e-12
    lb.lb_memcpy_h2d(g["f0"])
    for n in range(meta["nsteps"]):
        lb.lb_collide_fe(hy, meta["a"], meta["b"], meta["kappa"], phi, grad, delsq)
        lb.lb_halo()
        lb.lb_propagation()
        # deferred modes: the propagation stays pending for the next collision
        assert lb.state()[1] == (0 if mode == 0 else 1)
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(g["f_final"], 1)) < 1e-12
    lb.free()


@pytest.mark.parametrize("mode", [0, 3, 1], ids=["eager", "fused_halo", "fused"])
def test_binary_steps_on_the_slab_path(mode):
    """ndist = 2 with the X halo of both distributions and of phi through a
    1-rank RCCL ring = the single-rank run, bit for bit."""
    import ludwig_amd
    import torch
    g = load_golden("bin_q19_b")
    meta = g["meta"]
    out = []
    for ring in (False, True):
        lb = _lb(meta, mode=mode)
        if ring:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.fe_scheme_set(27, 1)
        phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        lb.lb_memcpy_h2d(g["f0"])
        for _ in range(meta["nsteps"]):
            lb.phi_to_field(phi)
            lb.field_halo_n(phi, 1)
            lb.field_grad(phi, grad, delsq)
            lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"],
                                 meta["mobility"], phi, grad, delsq)
            lb.lb_halo()
            lb.lb_propagation()
        out.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.free()
    assert np.array_equal(out[0], out[1])
    assert relmax(out[1], interior(g["f_final"], 1)) < 1e-12


def test_binary_fused_is_eager_at_every_observation():
    """FUSED with two distributions on one GPU (halo swap and propagation of
    both pending between steps): copies out after any call of the step and phi
    taken from the pending state equal EAGER bit for bit."""
    import ludwig_amd
    import torch
    g = load_golden("bin_q19_a")
    meta = g["meta"]
    obs = []
    for mode in (0, 1):
        rec = []
        lb = _lb(meta, mode=mode)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.fe_scheme_set(7, 1)
        phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        lb.lb_memcpy_h2d(g["f0"])
        for n in range(5):
            lb.phi_to_field(phi)
            rec.append(interior(_host(lb, phi), 1).copy())
            lb.field_halo_n(phi, 1)
            lb.field_grad(phi, grad, delsq)
            lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"],
                                 meta["mobility"], phi, grad, delsq)
            if n == 1:
                rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
            lb.lb_halo()
            if n == 2:
                rec.append(lb.lb_memcpy_d2h().copy())       # the halo of both included
            lb.lb_propagation()
            if n == 3:
                rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())   # flushes
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.free()
        obs.append(rec)
    for a, b in zip(obs[0], obs[1]):
        assert np.array_equal(a, b)


def test_binary_fused_halo_is_eager_at_every_observation():
    """FUSED_HALO with two distributions: copies out after any call of the
    step, the wall bounce-back between lb_halo and lb_propagation, and phi
    taken while the propagation is pending, all equal EAGER bit for bit."""
    import ludwig_amd
    import torch
    g = load_golden("bin_q19_a")
    meta = g["meta"]
    nv = meta["nvel"]
    obs = []
    for mode in (0, 3):
        rec = []
        lb = _lb(meta, mode=mode)
        hy = ludwig_amd.Hydro(lb.nall, lb.device,
                              status=np.zeros(lb.nall, dtype=np.int8))
        torch.cuda.synchronize()
        lb.wall_map((0, 0, 1), hy.status)
        lb.wall_links_build(hy.status, (0, 0, 1))
        lb.wall_velocity_set((0.01, 0, 0), (-0.01, 0, 0))
        lb.fe_scheme_set(7, 1)
        phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        lb.lb_memcpy_h2d(g["f0"])
        for n in range(5):
            lb.phi_to_field(phi)
            rec.append(interior(_host(lb, phi), 1).copy())
            lb.field_halo_n(phi, 1)
            lb.field_grad(phi, grad, delsq)
            lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"],
                                 meta["mobility"], phi, grad, delsq)
            if n == 1:
                rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
            lb.lb_halo()
            lb.wall_bbl()
            if n == 2:
                rec.append(lb.lb_memcpy_d2h().copy())       # halo and bounce included
            lb.lb_propagation()
            if n == 3:
                rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())   # flushes
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        rec.append(lb.wall_momentum())
        lb.free()
        obs.append(rec)
    assert len(obs[0]) == len(obs[1])
    for a, b in zip(obs[0], obs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("which", ["binary", "relax"])
def test_collisions_with_strided_gradient_and_hydro_arrays(which):
    """lbmi_fe_symm_t::nsite and lbmi_hydro_t::nsite: grad phi, force and u
    whose components lie further apart than the lattice's nsite (fields with
    Lees-Edwards buffer planes, field.c / hydro.c): the reference's results,
    the padding untouched."""
    import ludwig_amd
    import torch
    from tests.common import golden_relax_names
    name = "bin_q19_a" if which == "binary" else golden_relax_names()[0]
    g = load_golden(name)
    meta = g["meta"]
    nv = meta["nvel"]
    lb = _lb(meta) if which == "binary" else None
    if lb is None:
        lb = ludwig_amd.LB(nv, tuple(meta["nlocal"]), 1)
        lb.relaxation_set("m10", meta["eta"], meta["zeta"])
        lb.body_force_set(meta["fbody"])
    nsite = int(np.prod(lb.nall))
    stride = nsite + 3 * lb.nall[1] * lb.nall[2] + 5
    marker = -3.5
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    hy.u = torch.full((3, stride), marker, dtype=torch.float64, device=lb.device)
    hy.stride = stride
    gpad = np.full((3, stride), 1e30)                  # poison between the components
    gpad[:, :nsite] = g["grad"].reshape(3, nsite)
    grad = torch.from_numpy(gpad).to(lb.device)
    torch.cuda.synchronize(lb.device)
    lb.lb_memcpy_h2d(g["f0"])
    if which == "binary":
        lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                             _dev(lb, g["phi"]), grad, _dev(lb, g["delsq"]), grad_stride=stride)
    else:
        lb.lb_collide_fe(hy, meta["a"], meta["b"], meta["kappa"], _dev(lb, g["phi"]), grad,
                         _dev(lb, g["delsq"]), grad_stride=stride)
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, 1), interior(g["f_collide"], 1)) < 1e-12
    u = _host(lb, hy.u)
    assert relmax(interior(u[:, :nsite].reshape((3,) + lb.nall), 1), interior(g["u"], 1)) < 1e-12
    assert np.all(u[:, nsite:] == marker)
    lb.free()


@pytest.mark.parametrize("nvel", [19, 27])
def test_two_distributions_in_the_blocked_order(nvel):
    """FUSED on one GPU with two distributions: the deferred state is kept in
    the blocked order [site/256][n*nvel + p][site%256] (lbmi_lb_state says
    so), phi_lb_to_field reads it there, a flush converts back. Bit for bit
    what the same handle gives with lbmi_tune blocked 0, observers included."""
    import ludwig_amd
    import torch
    nlocal = (20, 12, 16)
    a, b, kappa, mob = -0.00625, 0.00625, 0.004, 1.25
    p = lbo.make_param(nvel, nlocal, 1, "m10", 0.1, 0.2, 1.0, (1e-6, 0, 0))
    rng = np.random.default_rng(23)
    f2 = np.zeros((2 * nvel,) + lbo.nall(p))
    f2[:nvel] = lbo.init_synthetic(p)
    w = lbo.model(nvel)["wv"]
    ph0 = 0.3 * rng.standard_normal(nlocal)
    for q in range(nvel):
        interior(f2[nvel + q], 1)[...] = w[q] * ph0 * (1 + 0.01 * rng.standard_normal(nlocal))
    out = []
    for blocked in (0, 1):
        lb = ludwig_amd.LB(nvel, nlocal, 1, ndist=2, mode=ludwig_amd.FUSED)
        lb.relaxation_set("m10", 0.1, 0.2)
        lb.tune("blocked", blocked)
        lb.fe_scheme_set(7, 1)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
        delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
        lb.lb_memcpy_h2d(f2)
        rec, orders = [], []
        for n in range(6):
            lb.phi_to_field(phi)
            rec.append(interior(_host(lb, phi), 1).copy())
            lb.field_halo_n(phi, 1)
            lb.field_grad(phi, grad, delsq)
            lb.lb_collide_binary(hy, a, b, kappa, mob, phi, grad, delsq)
            orders.append(lb.state()[2])
            lb.lb_halo()
            lb.lb_propagation()
            if n == 3:
                rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())      # flush: back to SoA
                assert tuple(lb.state()) == (0, 0, 0)
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        rec.append(interior(_host(lb, hy.u), 1).copy())
        lb.free()
        out.append((rec, orders))
    # SoA throughout / blocked from the second collision on (the first is in
    # place), and again after the flush
    assert out[0][1] == [0] * 6
    assert out[1][1] == [0, 1, 1, 1, 0, 1]
    for x, y in zip(out[0][0], out[1][0]):
        assert np.array_equal(x, y)
