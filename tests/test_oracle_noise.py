"""The oracle's isothermal fluctuations (oracle/lb_oracle.c: lbo_collide_noise,
restating collision.c:476-518, 1745-1920 and noise.c:397-487) against the
compiled reference: fixtures noise_q19_* of oracle/make_golden.py (ref_driver
dump with noise->on[NOISE_RHO] and a temperature), which hold the reference's
generator states before the first and after the last step. The generator is
integer arithmetic: its states must come out bit for bit. CPU only."""

import glob
import os

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import GOLDEN_DIR, interior, load_golden, relmax, status_from_meta

TOL = 5.0e-15


def noise_names():
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "noise_q19_*.npz")))


def _param(meta):
    return lbo.make_param(meta["nvel"], meta["nlocal"], meta["nhalo"], meta["scheme"],
                          meta["eta"], meta["zeta"], meta["rho0"], meta["fbody"])


def test_fixtures_exist():
    assert len(noise_names()) == 4


@pytest.mark.parametrize("name", noise_names())
def test_first_collision(name):
    g = load_golden(name)
    meta = g["meta"]
    assert meta["kt"] > 0.0
    p = _param(meta)
    h = meta["nhalo"]
    f = g["f0"].copy()
    rho = np.zeros(f.shape[1:])
    u = np.zeros((3,) + f.shape[1:])
    st = status_from_meta(meta)
    state = g["noise0"].copy()
    eta = g["eta"].copy() if meta["visc"] else None
    lbo.collide_noise(p, f, g["force"].copy(), st, state, meta["kt"], meta["ghosts"],
                      eta=eta, rho=rho, u=u)
    assert relmax(interior(f, h), interior(g["f_collide"], h)) < TOL
    assert relmax(interior(rho, h), interior(g["rho"], h)) < TOL
    assert relmax(interior(u, h), interior(g["u"], h)) < TOL
    # the fluctuations are there: the same collision without them differs
    f2 = g["f0"].copy()
    if meta["visc"]:
        lbo.collide_visc(p, f2, g["force"].copy(), st, eta)
    else:
        lbo.collide(p, f2, g["force"].copy(), st)
    assert relmax(interior(f2, h), interior(g["f_collide"], h)) > 1e-6
    # and they carry neither mass nor momentum (stress and ghost modes only)
    cv = lbo.model(19)["cv"].astype(float)
    d = interior(f, h) - interior(f2, h)
    assert np.max(np.abs(d.sum(axis=0))) < 1e-15
    assert np.max(np.abs(np.tensordot(cv.T, d, axes=(1, 0)))) < 1e-15
    if meta["solid"]:
        solid = interior(st, h) != 0
        # solid sites neither collide nor draw (collision.c:484-491)
        assert np.array_equal(interior(state, h)[:, solid], interior(g["noise0"], h)[:, solid])
        assert not np.array_equal(interior(state, h)[:, ~solid], interior(g["noise0"], h)[:, ~solid])


@pytest.mark.parametrize("name", noise_names())
def test_steps_and_generator_states(name):
    g = load_golden(name)
    meta = g["meta"]
    p = _param(meta)
    h = meta["nhalo"]
    f = g["f0"].copy()
    fp = np.zeros_like(f)
    st = status_from_meta(meta)
    state = g["noise0"].copy()
    eta = g["eta"].copy() if meta["visc"] else None
    for _ in range(meta["nsteps"]):
        lbo.collide_noise(p, f, g["force"].copy(), st, state, meta["kt"], meta["ghosts"], eta=eta)
        lbo.halo(p, f)
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    assert relmax(interior(f, h), interior(g["f_final"], h)) < 1e-13
    # integer arithmetic: bit for bit (two draws per fluid site and step with
    # ghost modes on, one without)
    assert np.array_equal(interior(state, h), interior(g["noise_final"], h))


def test_d3q27_is_refused():
    p = lbo.make_param(27, (4, 4, 4), 1, "m10", 0.1, 0.3, 1.0, (0.0, 0.0, 0.0))
    f = lbo.init_synthetic(p)
    state = np.ones((4,) + f.shape[1:], dtype=np.uint32)
    with pytest.raises(ValueError):
        lbo.collide_noise(p, f, None, None, state, 1e-4)


# --- the two-distribution collision with fluctuations (collision.c:884-900) ---

def test_binary_collision_and_steps_with_fluctuations():
    g = load_golden("noise_bin_q19_a")
    meta = g["meta"]
    assert meta["kt"] > 0.0 and meta["ndist"] == 2
    p = lbo.make_param(19, meta["nlocal"], 1, "m10", meta["eta"], meta["zeta"], 1.0, meta["fbody"])
    nv = 19
    f2 = np.ascontiguousarray(g["f0"]).copy()
    u = np.zeros((3,) + f2.shape[1:])
    state = g["noise0"].copy()
    lbo.collide_binary_noise(p, f2, None, meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                             np.ascontiguousarray(g["phi"]), np.ascontiguousarray(g["grad"]),
                             np.ascontiguousarray(g["delsq"]), state, meta["kt"], True, u)
    assert relmax(interior(f2[:nv], 1), interior(g["f_collide"][:nv], 1)) < TOL
    assert relmax(interior(f2[nv:], 1), interior(g["f_collide"][nv:], 1)) < TOL
    assert relmax(interior(u, 1), interior(g["u"], 1)) < TOL
    # whole steps; the generator states bit for bit (two draws per site and step)
    f2 = np.ascontiguousarray(g["f0"]).copy()
    fp2 = np.zeros_like(f2)
    state = g["noise0"].copy()
    for _ in range(meta["nsteps"]):
        phi = lbo.phi_from_g(p, f2)
        lbo.field_halo(p, phi, 1)
        gr, d2 = lbo.grad(p, phi, 27)
        lbo.collide_binary_noise(p, f2, None, meta["a"], meta["b"], meta["kappa"],
                                 meta["mobility"], phi, gr, d2, state, meta["kt"], True)
        lbo.halo(p, f2)
        lbo.propagate(p, f2[:nv], fp2[:nv])
        lbo.propagate(p, f2[nv:], fp2[nv:])
        f2, fp2 = fp2, f2
    assert relmax(interior(f2, 1), interior(g["f_final"], 1)) < 1e-13
    assert np.array_equal(interior(state, 1), interior(g["noise_final"], 1))
