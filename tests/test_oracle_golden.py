"""Pin the oracle (oracle/lb_oracle.c) against the compiled reference.

The fixtures in tests/golden were produced by oracle/make_golden.py from the
reference itself (oracle/_ref). CPU only.
"""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import (RTOL_CONSERVED, golden_names, interior, load_golden,
                          momentum_scale, relmax, shell1, status_from_meta,
                          xplanes)

# The oracle follows the reference's summation order with FMA contraction
# off; agreement is at rounding level (the d3q19 reference uses literal
# coefficients, e.g. 1.0/72.0, where the oracle forms wv*na*ma).
TOL = 5.0e-15


def param_from_meta(meta):
    return lbo.make_param(meta["nvel"], meta["nlocal"], meta["nhalo"],
                          meta["scheme"], meta["eta"], meta["zeta"],
                          meta["rho0"], meta["fbody"])


@pytest.mark.parametrize("name", golden_names())
def test_model_tables_consistent(name):
    g = load_golden(name)
    m = lbo.model(g["meta"]["nvel"])
    nvel = m["nvel"]
    # reference tests/unit/test_lb_model.c:103-375
    assert np.all(m["cv"][0] == 0)
    assert abs(m["wv"].sum() - 1.0) < 1e-15
    ortho = (m["ma"] * m["wv"][None, :]) @ m["ma"].T * m["na"][:, None]
    assert np.max(np.abs(ortho - np.eye(nvel))) < 5 * np.finfo(float).eps * 10
    assert np.max(np.abs(m["mi"] @ m["ma"] - np.eye(nvel))) < 1e-14


@pytest.mark.parametrize("name", golden_names())
def test_synthetic_init_matches_reference_driver(name):
    g = load_golden(name)
    p = param_from_meta(g["meta"])
    f = lbo.init_synthetic(p)
    h = g["meta"]["nhalo"]
    assert relmax(interior(f, h), interior(g["f0"], h)) < 1e-15


@pytest.mark.parametrize("name", golden_names())
def test_collide(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param_from_meta(meta)
    h = meta["nhalo"]
    f = g["f0"].copy()
    rho = np.zeros(f.shape[1:])
    u = np.zeros((3,) + f.shape[1:])
    st = status_from_meta(meta)
    lbo.collide(p, f, g["force"].copy(), st, rho, u)
    assert relmax(interior(f, h), interior(g["f_collide"], h)) < TOL
    assert relmax(interior(rho, h), interior(g["rho"], h)) < TOL
    assert relmax(interior(u, h), interior(g["u"], h)) < TOL
    if meta["solid"]:
        solid = interior(st, h) != 0
        # non-fluid sites untouched (collision.c:299-304)
        assert np.array_equal(interior(f, h)[:, solid],
                              interior(g["f0"], h)[:, solid])


@pytest.mark.parametrize("name", [n for n in golden_names()
                                  if "f_halo" in load_golden(n)])
def test_halo_exact(name):
    g = load_golden(name)
    p = param_from_meta(g["meta"])
    f = g["f_collide"].copy()
    lbo.halo(p, f)
    # pure copies: bit-exact over interior + the exchanged width-1 shell
    h = g["meta"]["nhalo"]
    assert np.array_equal(shell1(f, h), shell1(g["f_halo"], h))


@pytest.mark.parametrize("name", golden_names())
def test_propagate_exact(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param_from_meta(meta)
    h = meta["nhalo"]
    f = g["f_collide"].copy()
    lbo.halo(p, f)
    fp = np.zeros_like(f)
    lbo.propagate(p, f, fp)
    # pure copies: bit-exact in every x-interior plane (y/z halo included)
    assert np.array_equal(xplanes(fp, h), xplanes(g["f_prop"], h))


@pytest.mark.parametrize("name", golden_names())
def test_multi_step(name):
    g = load_golden(name)
    meta = g["meta"]
    p = param_from_meta(meta)
    h = meta["nhalo"]
    f = g["f0"].copy()
    fp = np.zeros_like(f)
    st = status_from_meta(meta)
    force = g["force"].copy()
    for _ in range(meta["nsteps"]):
        f, fp = lbo.step(p, f, fp, force, st)
    assert relmax(interior(f, h), interior(g["f_final"], h)) < 10 * TOL
    mo = lbo.moments(p, f, st)
    mg = lbo.moments(p, np.ascontiguousarray(g["f_final"]), st)
    assert abs(mo[1] - mg[1]) / mg[1] < RTOL_CONSERVED
    gscale = momentum_scale(g["f_final"], lbo.model(meta["nvel"])["cv"], h)
    assert np.max(np.abs(mo[5:8] - mg[5:8])) / gscale < RTOL_CONSERVED


def test_trt_d3q27_rejected():
    # The reference reads uninitialised ghost rates for d3q27 + trt
    # (collision.c:1487-1534); we reject the combination.
    p = lbo.make_param(27, (4, 4, 4), 1, "trt")
    f = lbo.init_synthetic(p)
    with pytest.raises(ValueError):
        lbo.collide(p, f)


@pytest.mark.parametrize("name", [n for n in golden_names()
                                  if "records" in load_golden(n)])
def test_record_stream(name):
    # lb_io_aggr_pack / lb_write_buf, model.c:1385-1510
    g = load_golden(name)
    p = param_from_meta(g["meta"])
    f = np.ascontiguousarray(np.nan_to_num(g["f_final"]))
    rec = lbo.records_pack(p, f)
    assert np.array_equal(rec, g["records"])
    f2 = np.zeros_like(f)
    lbo.records_unpack(p, f2, rec)
    h = g["meta"]["nhalo"]
    assert np.array_equal(interior(f2, h), interior(f, h))
    # halo sites untouched
    assert np.count_nonzero(f2) == np.count_nonzero(interior(f2, h))


# --- viscosity model: local relaxation times from hydro->eta ------------------

from tests.common import golden_visc_names  # noqa: E402


@pytest.mark.parametrize("name", golden_visc_names())
def test_collision_with_local_viscosity(name):
    g = load_golden(name)
    meta = g["meta"]
    p = lbo.make_param(meta["nvel"], meta["nlocal"], meta["nhalo"],
                       meta["scheme_name"], meta["eta"], meta["zeta"],
                       meta["rho0"], meta["fbody"])
    eta = np.ascontiguousarray(g["eta"])
    force = np.ascontiguousarray(g["force"])
    f = np.ascontiguousarray(g["f0"]).copy()
    rho = np.zeros(f.shape[1:])
    u = np.zeros((3,) + f.shape[1:])
    lbo.collide_visc(p, f, force, None, eta, rho, u)
    h = meta["nhalo"]
    assert relmax(interior(f, h), interior(g["f_collide"], h)) < 5e-15
    assert relmax(interior(u, h), interior(g["u"], h)) < 5e-15
    # it is not the constant-viscosity collision
    f1 = np.ascontiguousarray(g["f0"]).copy()
    lbo.collide(p, f1, force)
    assert relmax(interior(f1, h), interior(g["f_collide"], h)) > 1e-7
    f = np.ascontiguousarray(g["f0"]).copy()
    fp = np.zeros_like(f)
    for _ in range(meta["nsteps"]):
        lbo.collide_visc(p, f, force, None, eta)
        lbo.halo(p, f)
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    assert relmax(interior(f, h), interior(g["f_final"], h)) < 1e-13
