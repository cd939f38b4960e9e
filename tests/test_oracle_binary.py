"""Pin the oracle's two-distribution (symmetric_lb) step -- row f4:
phi_lb_to_field, lb_collision_binary (collision.c:610-1027), lb_halo and
lb_propagation of both distributions -- against the compiled reference. CPU."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import golden_binary_names, interior, load_golden, relmax


def param(meta):
    return lbo.make_param(meta["nvel"], meta["nlocal"], 1, "m10", meta["eta"],
                          meta["zeta"], 1.0, meta["fbody"])


@pytest.mark.parametrize("name", golden_binary_names())
def test_phi_from_g_and_gradients(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param(meta)
    phi = lbo.phi_from_g(p, np.ascontiguousarray(g["f0"]))
    assert np.array_equal(interior(phi, 1), interior(g["phi"], 1))
    lbo.field_halo(p, phi, 1)
    assert np.array_equal(phi, g["phi"])
    grad, delsq = lbo.grad(p, phi, 27)
    assert np.array_equal(interior(grad, 1), interior(g["grad"], 1))
    assert np.array_equal(interior(delsq, 1), interior(g["delsq"], 1))


@pytest.mark.parametrize("name", golden_binary_names())
def test_binary_collision(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param(meta)
    f2 = np.ascontiguousarray(g["f0"]).copy()
    u = np.zeros((3,) + f2.shape[1:])
    lbo.collide_binary(p, f2, None, meta["a"], meta["b"], meta["kappa"],
                       meta["mobility"], np.ascontiguousarray(g["phi"]),
                       np.ascontiguousarray(g["grad"]),
                       np.ascontiguousarray(g["delsq"]), u)
    nv = meta["nvel"]
    assert relmax(interior(f2[:nv], 1), interior(g["f_collide"][:nv], 1)) < 5e-15
    assert relmax(interior(f2[nv:], 1), interior(g["f_collide"][nv:], 1)) < 5e-15
    assert relmax(interior(u, 1), interior(g["u"], 1)) < 5e-15
    # the re-projection puts the order parameter back exactly: sum_p g_p = phi
    assert relmax(interior(f2[nv:].sum(axis=0), 1), interior(g["phi"], 1)) < 1e-14


@pytest.mark.parametrize("name", golden_binary_names())
def test_binary_steps(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param(meta)
    f2 = np.ascontiguousarray(g["f0"]).copy()
    fp2 = np.zeros_like(f2)
    for _ in range(meta["nsteps"]):
        f2, fp2, _, _, _ = lbo.step_binary(p, f2, fp2, meta["a"], meta["b"],
                                           meta["kappa"], meta["mobility"])
    assert relmax(interior(f2, 1), interior(g["f_final"], 1)) < 1e-13
    # both conserved: total mass and total order parameter
    nv = meta["nvel"]
    for sl in (slice(0, nv), slice(nv, 2 * nv)):
        m0 = interior(g["f0"][sl], 1).sum()
        assert abs(interior(f2[sl], 1).sum() - m0) < 1e-12 * max(1.0, abs(m0))


# --- one distribution, fe->use_stress_relaxation (collision.c:413-429) -------

from tests.common import golden_relax_names  # noqa: E402


@pytest.mark.parametrize("name", golden_relax_names())
def test_stress_relaxation_collision_and_steps(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param(meta)
    phi = np.ascontiguousarray(g["phi"])
    grad = np.ascontiguousarray(g["grad"])
    delsq = np.ascontiguousarray(g["delsq"])
    f = np.ascontiguousarray(g["f0"]).copy()
    rho = np.zeros(f.shape[1:])
    u = np.zeros((3,) + f.shape[1:])
    lbo.collide_fe(p, f, None, None, meta["a"], meta["b"], meta["kappa"], phi,
                   grad, delsq, rho, u)
    assert relmax(interior(f, 1), interior(g["f_collide"], 1)) < 5e-15
    assert relmax(interior(rho, 1), interior(g["rho"], 1)) < 5e-15
    assert relmax(interior(u, 1), interior(g["u"], 1)) < 5e-15
    # the stress changes the result: this is not the plain collision
    f1 = np.ascontiguousarray(g["f0"]).copy()
    lbo.collide(p, f1)
    assert relmax(interior(f1, 1), interior(g["f_collide"], 1)) > 1e-6
    # whole steps with phi fixed
    f = np.ascontiguousarray(g["f0"]).copy()
    fp = np.zeros_like(f)
    for _ in range(meta["nsteps"]):
        lbo.collide_fe(p, f, None, None, meta["a"], meta["b"], meta["kappa"],
                       phi, grad, delsq)
        lbo.halo(p, f)
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    assert relmax(interior(f, 1), interior(g["f_final"], 1)) < 1e-13
