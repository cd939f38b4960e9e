"""Pin the oracle's flat walls and bounce-back on links (row f4: wall_init_map,
wall_init_boundaries, wall_init_uw, wall_bbl of wall.c) against the compiled
reference. CPU."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import (golden_slip_names, golden_wall_names, interior,
                          load_golden, relmax)


def setup(g):
    meta = g["meta"]
    p = lbo.make_param(meta["nvel"], meta["nlocal"], 1, "m10", meta["eta"],
                       meta["zeta"])
    return meta, p


@pytest.mark.parametrize("name", golden_wall_names())
def test_wall_map_and_links_exact(name):
    g = load_golden(name)
    meta, p = setup(g)
    status = np.zeros(lbo.nall(p), dtype=np.int8)
    if meta["solid"] == 1:
        status[2:4, 2:4, 2:4] = 1               # the block of the driver
    lbo.wall_map(p, meta["isboundary"], status)
    # (solid = 2: MAP_COLLOID marks put on AFTER the links were built)
    assert np.array_equal(status, np.where(g["status"] == 2, 0, g["status"]))
    li, lj, lp, lu = lbo.wall_links(p, status, meta["isboundary"])
    assert len(li) == meta["nlink"]
    for mine, ref in ((li, "linki"), (lj, "linkj"), (lp, "linkp"), (lu, "linku")):
        assert np.array_equal(mine, g[ref])      # same links, same order


@pytest.mark.parametrize("name", golden_wall_names())
def test_wall_steps(name):
    g = load_golden(name)
    meta, p = setup(g)
    status = np.ascontiguousarray(g["status"])
    links = (g["linki"], g["linkj"], g["linkp"], g["linku"])
    f = np.ascontiguousarray(g["f0"]).copy()
    fp = np.zeros_like(f)
    fnet = np.zeros(3)
    for n in range(meta["nsteps"]):
        lbo.collide(p, f, None, status)
        lbo.halo(p, f)
        lbo.wall_bbl(p, f, links, meta["ubot"], meta["utop"], fnet, status)
        if n == 0:
            # the solid-side entries the bounce-back wrote, and the fluid
            for k in range(len(links[0])):
                q = meta["nvel"] - links[2][k]
                assert abs(f.reshape(meta["nvel"], -1)[q, links[1][k]]
                           - g["f_bbl"].reshape(meta["nvel"], -1)[q, links[1][k]]) < 1e-15
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(f, 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-13
    scale = np.abs(np.array(meta["fnet"])).max()
    assert np.max(np.abs(fnet - np.array(meta["fnet"]))) < 1e-12 * max(scale, 1.0)
    if meta["solid"] == 2:
        assert (status.reshape(-1)[links[0]] == 2).any()    # the branch is taken
    # no-slip walls at rest conserve mass exactly
    if meta["ubot"][1] == 0.0 and meta["utop"][1] == 0.0:
        assert abs(interior(f, 1)[:, fl].sum() - interior(g["f0"], 1)[:, fl].sum()) < 1e-11


# Partial slip (wall_init_boundaries_slip, wall_bbl_slip_kernel)

def test_slip_table():
    active, s = lbo.wall_slip_table((0.5, 0.25, 1.0), (0.0, 0.75, 0.3))
    assert active
    # wall_slip (wall.c:285-316): faces, then edges = mean of the two faces
    assert list(s[:7]) == [0.0, 0.5, 0.0, 0.25, 0.75, 1.0, 0.3]
    assert s[7] == 0.5 * (0.5 + 0.25) and s[8] == 0.5 * (0.5 + 0.75)      # XB_YB, XB_YT
    assert s[9] == 0.5 * (0.5 + 1.0) and s[10] == 0.5 * (0.5 + 0.3)       # XB_ZB, XB_ZT
    assert s[11] == 0.5 * (0.0 + 0.25) and s[14] == 0.5 * (0.0 + 0.3)     # XT_YB, XT_ZT
    assert s[15] == 0.5 * (0.25 + 1.0) and s[18] == 0.5 * (0.75 + 0.3)    # YB_ZB, YT_ZT
    assert not lbo.wall_slip_table((0, 0, 0), (0, 0, 0))[0]


@pytest.mark.parametrize("name", golden_slip_names())
def test_slip_links_exact(name):
    g = load_golden(name)
    meta, p = setup(g)
    status = lbo.wall_map(p, meta["isboundary"])
    assert np.array_equal(status, np.where(g["status"] == 2, 0, g["status"]))
    links = lbo.wall_links(p, status, meta["isboundary"])
    for mine, ref in zip(links, ("linki", "linkj", "linkp", "linku")):
        assert np.array_equal(mine, g[ref])
    lk, lq, ls = lbo.wall_slip_links(p, status, links)
    assert np.array_equal(lk, g["linkk"])
    assert np.array_equal(lq, g["linkq"])
    assert np.array_equal(ls, g["links"])
    # the fixtures exercise faces and, where a velocity has a component along
    # an edge (three non-zero components: D3Q27 only), edges
    assert (ls > 0).any() and (ls == 0).any()
    if sum(meta["isboundary"]) > 1 and meta["nvel"] == 27:
        assert (ls >= 7).any()


@pytest.mark.parametrize("name", golden_slip_names())
def test_slip_steps(name):
    g = load_golden(name)
    meta, p = setup(g)
    status = np.ascontiguousarray(g["status"])
    links = (g["linki"], g["linkj"], g["linkp"], g["linku"])
    slinks = (g["linkk"], g["linkq"], g["links"])
    active, stab = lbo.wall_slip_table(meta["sbot"], meta["stop"])
    assert active
    nvel = meta["nvel"]
    f = np.ascontiguousarray(g["f0"]).copy()
    fp = np.zeros_like(f)
    fnet = np.zeros(3)
    for n in range(meta["nsteps"]):
        lbo.collide(p, f, None, status)
        lbo.halo(p, f)
        lbo.wall_bbl_slip(p, f, links, slinks, stab, fnet, status)
        if n == 0:
            mine = f.reshape(nvel, -1)[nvel - links[2], links[1]]
            ref = g["f_bbl"].reshape(nvel, -1)[nvel - links[2], links[1]]
            assert np.max(np.abs(mine - ref)) < 1e-15
        lbo.propagate(p, f, fp)
        f, fp = fp, f
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(f, 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-13
    scale = np.abs(np.array(meta["fnet"])).max()
    assert np.max(np.abs(fnet - np.array(meta["fnet"]))) < 1e-12 * max(scale, 1.0)
    if meta["solid"] == 2:
        assert (status.reshape(-1)[links[0]] == 2).any()    # the branch is taken
        return
    # slip links come in pairs with one s: mass is conserved
    assert abs(interior(f, 1)[:, fl].sum() - interior(g["f0"], 1)[:, fl].sum()) < 1e-11
