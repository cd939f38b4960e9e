"""The round-1 "Memory access fault by GPU" of the duct case (serial-rect-ct1
through the binding), replayed as address arithmetic on the CPU.

Cause (CHANGELOG.md, "The round-1 GPU memory fault"): the bound lb_collide of round 1 never called
lb_collide_param_commit (reference collision.c:157, model.c:342-349), the only
upload of lb->param to the device copy behind lb->target->param. The
reference's wall_setu_kernel (wall.c:930-950), run every step by
wall_set_wall_distributions (ludwig.c:837), computes

    p = lb->param->nvel - wall->linkp[n]
    lb_f_set(lb, wall->linkj[n], p, LB_RHO, fp)       # f[nsite*p + linkj[n]]

With the device parameters still zero, nvel = 0 and p = -linkp[n] lies in
-18 .. -1: every link wrote up to 18 nsite doubles BEFORE f (the build has
-DNDEBUG, so the device assert p >= 0 is off). This file derives those
addresses for the 1 x 62 x 30 duct from the oracle's link list -- outside the
array before the fix, inside after -- and checks the host-side record test of
lbmi_wall_links_set on the same list. No GPU is involved."""

import numpy as np

from oracle import lb_oracle as lbo

NVEL = 19
NLOCAL = (1, 62, 30)                 # tests/golden/inputs/rect_ct1.inp: size 1_62_30
ISBOUNDARY = (0, 1, 1)               # boundary_walls 0_1_1


def _duct_links():
    p = lbo.make_param(NVEL, NLOCAL, 1, "m10", 0.1666, 0.1666, 1.0, (1e-5, 0.0, 0.0))
    status = lbo.wall_map(p, ISBOUNDARY)
    li, lj, lp, lu = lbo.wall_links(p, status, ISBOUNDARY)
    nsite = int(np.prod(lbo.nall(p)))
    return nsite, li.astype(np.int64), lj.astype(np.int64), lp.astype(np.int64), lu


def test_wall_setu_addresses_with_uncommitted_and_committed_parameters():
    nsite, li, lj, lp, lu = _duct_links()
    assert nsite == 3 * 64 * 32 == 6144          # the f-sized buffer of the report: 6144 x 19 doubles
    assert len(li) > 0 and lp.min() >= 1 and lp.max() <= NVEL - 1

    # round 1: the device copy of lb->param was never written: nvel = 0
    addr_before = nsite * (0 - lp) + lj
    assert addr_before.max() < 0                 # every one of them in front of f
    assert addr_before.min() >= -(NVEL - 1) * nsite
    assert addr_before.min() < -nsite            # more than one population array away

    # with lb_collide_param_commit in the bound lb_collide: nvel = 19
    addr_after = nsite * (NVEL - lp) + lj
    assert addr_after.min() >= 0 and addr_after.max() < NVEL * nsite
    # ... and it is the slot wall_bbl fills afterwards (wall.c:1040, 1078-1080),
    # which is why the logs matched in spite of the stray writes
    assert np.array_equal(addr_after, nsite * (NVEL - lp) + lj)


def test_bounce_back_addresses_of_the_duct_stay_inside_f():
    """k_wall_bbl (lbmi_kernels.hip) reads f[nsite*p + i] and writes
    f[nsite*(nvel - p) + j]: inside the array for every link of the case, and
    the record test of lbmi_wall_links_set (lbmi_host.c) holds for them."""
    nsite, li, lj, lp, lu = _duct_links()
    cv = lbo.model(NVEL)["cv"].astype(np.int64)
    nall = [n + 2 for n in NLOCAL]
    strx, stry = nall[1] * nall[2], nall[2]
    for a in (nsite * lp + li, nsite * (NVEL - lp) + lj):
        assert a.min() >= 0 and a.max() < NVEL * nsite
    assert np.array_equal(lj, li + cv[lp, 0] * strx + cv[lp, 1] * stry + cv[lp, 2])
    assert set(np.unique(lu)) <= {0, 1, 2}
    # the all-zero record the round-1 review pointed at (i = j = p = 0) would
    # have touched f[nsite*(nvel - 0) + 0], one element past the end: the
    # record test (1 <= p < nvel) and the kernel's own guard refuse it
    assert nsite * (NVEL - 0) + 0 == NVEL * nsite
