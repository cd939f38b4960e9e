"""Pin the oracle's flat walls and bounce-back on links (row f4: wall_init_map,
wall_init_boundaries, wall_init_uw, wall_bbl of wall.c) against the compiled
reference. CPU."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import golden_wall_names, interior, load_golden, relmax


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
    if meta["solid"]:
        status[2:4, 2:4, 2:4] = 1               # the block of the driver
    lbo.wall_map(p, meta["isboundary"], status)
    assert np.array_equal(status, g["status"])
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
        lbo.wall_bbl(p, f, links, meta["ubot"], meta["utop"], fnet)
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
    # no-slip walls at rest conserve mass exactly
    if meta["ubot"][1] == 0.0 and meta["utop"][1] == 0.0:
        assert abs(interior(f, 1)[:, fl].sum() - interior(g["f0"], 1)[:, fl].sum()) < 1e-11
