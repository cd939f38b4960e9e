"""Pin the oracle's symmetric free-energy chain (row f2) against the
compiled reference: field_halo (width 2), grad_3d_7pt_fluid /
grad_3d_27pt_fluid, pth_stress_compute + pth_force_fluid_driver, and
phi_cahn_hilliard with advection of order 1..4. CPU only."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import golden_fe_names, interior, load_golden, relmax


def fe_param(meta):
    return lbo.make_param(19, meta["nlocal"], meta["nhalo"])


@pytest.mark.parametrize("name", golden_fe_names())
def test_field_halo_width2_exact(name):
    g = load_golden(name)
    meta = g["meta"]
    p = fe_param(meta)
    h = meta["nhalo"]
    phi = np.zeros_like(g["phi"])
    interior(phi, h)[...] = interior(g["phi"], h)
    lbo.field_halo(p, phi, 2)
    assert np.array_equal(phi, g["phi"])        # both halo layers, bit for bit


@pytest.mark.parametrize("name", golden_fe_names())
def test_gradient(name):
    """grad_3d_7pt_fluid / grad_3d_27pt_fluid, bit for bit (the oracle keeps
    the reference's summation order)."""
    g = load_golden(name)
    meta = g["meta"]
    p = fe_param(meta)
    grad, delsq = lbo.grad(p, np.ascontiguousarray(g["phi"]),
                           meta.get("grad_npt", 7))
    # computed region: interior + nextra = nhalo - 1 = 1 layer
    s = (slice(1, -1),) * 3
    assert np.array_equal(grad[(slice(None),) + s], g["grad"][(slice(None),) + s])
    assert np.array_equal(delsq[s], g["delsq"][s])


@pytest.mark.parametrize("name", golden_fe_names())
def test_symmetric_force(name):
    g = load_golden(name)
    meta = g["meta"]
    p = fe_param(meta)
    h = meta["nhalo"]
    force = np.zeros_like(g["force"])
    lbo.symm_force(p, meta["a"], meta["b"], meta["kappa"],
                   np.ascontiguousarray(g["phi"]),
                   np.ascontiguousarray(g["grad"]),
                   np.ascontiguousarray(g["delsq"]), force)
    assert relmax(interior(force, h), interior(g["force"], h)) < 1e-13
    # a stress divergence sums to zero over a periodic box
    total = interior(force, h).reshape(3, -1).sum(axis=1)
    assert np.max(np.abs(total)) < 1e-15


@pytest.mark.parametrize("name", golden_fe_names())
def test_u_halo_and_cahn_hilliard(name):
    g = load_golden(name)
    meta = g["meta"]
    p = fe_param(meta)
    h = meta["nhalo"]
    # hydro_u_halo: width-1 swap of the 3-component field (hydro.c:190)
    u = np.zeros_like(g["u"])
    interior(u, h)[...] = interior(g["u"], h)
    lbo.field_halo(p, u, 1)
    s1 = (slice(None), slice(1, -1), slice(1, -1), slice(1, -1))
    assert np.array_equal(u[s1], g["u"][s1])
    phi = np.ascontiguousarray(g["phi"]).copy()
    lbo.cahn_hilliard(p, meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                      phi, np.ascontiguousarray(g["delsq"]), u,
                      order=meta.get("advection_order", 1))
    assert relmax(interior(phi, h), interior(g["phi_new"], h)) < 1e-14
    # conservative: sum phi unchanged to rounding
    assert abs(interior(phi, h).sum() - interior(g["phi"], h).sum()) < 1e-13
