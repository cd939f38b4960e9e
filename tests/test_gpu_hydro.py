"""hydro->rho, u on demand (lbmi_tune "hydro_lazy") and force arrays known to
hold zeros (lbmi_hydro_field_set): what the reference's lb_collide stores
every step (collision.c:563-596) and reads every step (:329-333), delivered
only when somebody wants it -- same values to rounding, checked against the
oracle and against the eager run at every call point of a step."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                       # noqa: E402
from tests.common import interior, relmax                  # noqa: E402

NLOCAL = (7, 6, 9)
FBODY = (1e-5, -2e-5, 3e-6)


def _force_field(nall):
    rng = np.random.default_rng(11)
    return 1e-4 * (rng.random((3,) + tuple(nall)) - 0.5)


def _status(nall):
    st = np.zeros(nall, dtype=np.int8)
    st[3:5, 2:4, 4:6] = 1
    return st


def _oracle_run(nvel, nsteps, force, status):
    """f, rho, u after nsteps: rho, u as the collision of the LAST step stored them."""
    p = lbo.make_param(nvel, NLOCAL, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    rho = np.zeros(lbo.nall(p))
    u = np.zeros((3,) + tuple(lbo.nall(p)))
    for _ in range(nsteps):
        f, fp = lbo.step(p, f, fp, force=force, status=status, rho=rho, u=u)
    return p, f, rho, u


def _fluid(status, nall):
    m = np.ones(nall, dtype=bool) if status is None else (status == 0)
    return m[1:-1, 1:-1, 1:-1]


@pytest.mark.parametrize("blocked", [1, 0])
@pytest.mark.parametrize("mode", [1, 3], ids=["fused", "fused_halo"])
@pytest.mark.parametrize("nvel", [19, 27])
@pytest.mark.parametrize("with_force,with_status", [(False, False), (True, True)])
def test_lazy_hydro_at_every_call_point(nvel, mode, blocked, with_force, with_status):
    """Ask for rho, u (lbmi_lb_hydro_sync, or implicitly by a flush) after the
    collision, after the halo swap and after the propagation of the last step:
    the oracle's values of that collision, and the run goes on undisturbed."""
    import ludwig_amd
    import torch
    nsteps = 3
    nall = tuple(n + 2 for n in NLOCAL)
    force = _force_field(nall) if with_force else None
    status = _status(nall) if with_status else None
    p, fref, rho_ref, u_ref = _oracle_run(nvel, nsteps, force, status)
    _, fref2, _, _ = _oracle_run(nvel, nsteps + 2, force, status)
    f0 = lbo.init_synthetic(p)
    fl = _fluid(status, nall)
    for point in ("collide", "halo", "propagation"):
        for how in ("sync", "flush"):
            lb = ludwig_amd.LB(nvel, NLOCAL, 1, mode=mode)
            lb.relaxation_set("m10", 0.1, 0.3)
            lb.body_force_set(FBODY)
            lb.tune("blocked", blocked)
            lb.tune("hydro_lazy", 1)
            hy = ludwig_amd.Hydro(nall, lb.device, force=force, status=status)
            lb.lb_memcpy_h2d(f0)

            def ask():
                if how == "sync":
                    lb.hydro_sync()
                    lb.synchronize()
                else:
                    lb.lb_memcpy_d2h()               # flushes: settles the debt too

            for n in range(nsteps + 2):
                lb.lb_collide(hy)
                if n == nsteps - 1 and point == "collide":
                    ask()
                lb.lb_halo()
                if n == nsteps - 1 and point == "halo":
                    ask()
                lb.lb_propagation()
                if n == nsteps - 1 and point == "propagation":
                    ask()
                if n == nsteps - 1:
                    torch.cuda.synchronize()
                    rho = interior(hy.rho.cpu().numpy(), 1)
                    u = interior(hy.u.cpu().numpy(), 1)
                    assert relmax(rho[fl], interior(rho_ref, 1)[fl]) < 1e-12, (point, how)
                    assert relmax(u[:, fl], interior(u_ref, 1)[:, fl]) < 1e-12, (point, how)
            out = lb.lb_memcpy_d2h()
            assert relmax(interior(out, 1)[:, fl], interior(fref2, 1)[:, fl]) < 1e-12
            lb.free()


def test_lazy_hydro_stores_nothing_until_asked_and_eager_modes_ignore_it():
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    p, fref, rho_ref, u_ref = _oracle_run(19, 2, None, None)
    f0 = lbo.init_synthetic(p)
    for mode, lazy_effective in ((1, True), (3, True), (0, False), (2, False)):
        lb = ludwig_amd.LB(19, NLOCAL, 1, mode=mode)
        lb.relaxation_set("m10", 0.1, 0.3)
        lb.body_force_set(FBODY)
        lb.tune("hydro_lazy", 1)
        hy = ludwig_amd.Hydro(nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        for _ in range(2):
            lb.lb_collide(hy)
            lb.lb_halo()
            lb.lb_propagation()
        lb.synchronize()
        torch.cuda.synchronize()
        written = float(hy.rho.abs().max()) > 0.0
        assert written == (not lazy_effective), mode
        lb.hydro_sync()
        lb.synchronize()
        torch.cuda.synchronize()
        assert relmax(interior(hy.rho.cpu().numpy(), 1), interior(rho_ref, 1)) < 1e-12
        assert relmax(interior(hy.u.cpu().numpy(), 1), interior(u_ref, 1)) < 1e-12
        lb.free()


def test_lazy_hydro_through_lbmi_lb_run_and_switching_off():
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    p, fref, rho_ref, u_ref = _oracle_run(19, 9, None, None)
    f0 = lbo.init_synthetic(p)
    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=1)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    lb.tune("hydro_lazy", 1)
    hy = ludwig_amd.Hydro(nall, lb.device)
    lb.lb_memcpy_h2d(f0)
    lb.run(hy, 9)
    lb.tune("hydro_lazy", 0)                    # settles what is owed
    lb.synchronize()
    torch.cuda.synchronize()
    assert relmax(interior(hy.rho.cpu().numpy(), 1), interior(rho_ref, 1)) < 1e-12
    assert relmax(interior(hy.u.cpu().numpy(), 1), interior(u_ref, 1)) < 1e-12
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(fref, 1)) < 1e-12
    lb.free()


@pytest.mark.parametrize("mode", [0, 1, 3], ids=["eager", "fused", "fused_halo"])
def test_force_known_to_be_zero_is_not_read(mode):
    """hydro_f_zero through the library marks the array: the collision does
    not read it until someone reports a write. Shown by writing to it behind
    the library's back -- no effect until lbmi_hydro_field_dirty (the
    contract of include/lbmi.h), full effect after."""
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    force = _force_field(nall)
    _, f_noforce, _, _ = _oracle_run(19, 2, None, None)
    p, f_force, _, _ = _oracle_run(19, 2, force, None)
    f0 = lbo.init_synthetic(p)

    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    hy = ludwig_amd.Hydro(nall, lb.device, force=np.ones_like(force))
    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))              # hydro_f_zero
    lb.synchronize()
    torch.cuda.synchronize()
    assert float(hy.force.abs().max()) == 0.0
    # a foreign writer the Python mirror cannot see either (.data does not
    # share the tensor's version counter: what another library's kernel is)
    hy.force.data.copy_(torch.from_numpy(force))
    torch.cuda.synchronize()
    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))              # zeros over "zeros": no launch
    lb.synchronize()
    assert float(hy.force.abs().max()) > 0.0
    lb.lb_memcpy_h2d(f0)
    for _ in range(2):
        lb.step(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f_noforce, 1)) < 1e-12

    lb.hydro_field_dirty(hy.force)                             # ... who owns up
    lb.lb_memcpy_h2d(f0)
    for _ in range(2):
        lb.step(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f_force, 1)) < 1e-12

    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))              # now it does launch
    lb.synchronize()
    torch.cuda.synchronize()
    assert float(hy.force.abs().max()) == 0.0
    lb.free()


def test_the_python_mirror_reports_writes_of_torch_to_a_zeroed_array():
    """The library keeps "known to hold zeros" by device address and relies on
    every other writer to report (include/lbmi.h). torch does not: the mirror
    checks the version counter of every tensor it has zeroed before each
    collision, and that the tensor at that address is still the same one --
    an in-place write by torch, or a new tensor that the caching allocator
    put where a dead one was, is reported for it."""
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    force = _force_field(nall)
    p, f_force, _, _ = _oracle_run(19, 2, force, None)
    f0 = lbo.init_synthetic(p)

    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=1)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    hy = ludwig_amd.Hydro(nall, lb.device, force=np.zeros_like(force))
    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))
    hy.force.copy_(torch.from_numpy(force))          # torch writes, says nothing
    torch.cuda.synchronize()
    lb.lb_memcpy_h2d(f0)
    for _ in range(2):
        lb.step(hy)
    assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f_force, 1)) < 1e-12

    # a zeroed tensor dies, a new one lands on its address
    lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))
    lb.synchronize()
    ptr = hy.force.data_ptr()
    hy.force = None
    torch.cuda.synchronize()
    t = torch.from_numpy(np.ascontiguousarray(force)).to(lb.device)
    if t.data_ptr() == ptr:                          # (the allocator usually does)
        hy.force = t
        lb.lb_memcpy_h2d(f0)
        for _ in range(2):
            lb.run(hy, 1)
        assert relmax(interior(lb.lb_memcpy_d2h(), 1), interior(f_force, 1)) < 1e-12
    lb.free()


def test_lb_dirty_after_the_caller_rewrote_f():
    """lbmi_lb_dirty: refused while anything is deferred, and what a lazy
    collision still owed is dropped with the state it would have been formed
    from."""
    import ludwig_amd
    nall = tuple(n + 2 for n in NLOCAL)
    p, _, _, _ = _oracle_run(19, 1, None, None)
    f0 = lbo.init_synthetic(p)
    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=1)
    lb.relaxation_set("m10", 0.1, 0.3)
    hy = ludwig_amd.Hydro(nall, lb.device)
    lb.lb_memcpy_h2d(f0)
    lb.step(hy)
    with pytest.raises(Exception):
        lb.lb_dirty()                                # propagation pending
    lb.lb_flush()
    lb.lb_dirty()
    lb.free()


def test_reference_step_order_with_u_zero_and_f_zero_every_step():
    """The housekeeping of ludwig.c:537-860 around a lazy step: hydro_f_zero,
    hydro_u_zero, lb_collide, lb_halo, lb_propagation, every step. u zeroed
    before the collision is not owed any more; the statistics step at the end
    gets the u of the last collision, zeros at the solid sites."""
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    status = _status(nall)
    p, fref, rho_ref, u_ref = _oracle_run(19, 4, None, status)
    f0 = lbo.init_synthetic(p)
    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=3)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    lb.tune("hydro_lazy", 1)
    hy = ludwig_amd.Hydro(nall, lb.device, force=np.zeros((3,) + nall), status=status)
    hy.u.fill_(7.0)
    lb.lb_memcpy_h2d(f0)
    for _ in range(4):
        lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))
        lb.hydro_field_set(hy.u, (0.0, 0.0, 0.0))
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.lb_propagation()
    lb.hydro_sync()
    lb.synchronize()
    torch.cuda.synchronize()
    u = hy.u.cpu().numpy()
    fl = status == 0
    assert relmax(interior(u, 1)[:, fl[1:-1, 1:-1, 1:-1]],
                  interior(u_ref, 1)[:, fl[1:-1, 1:-1, 1:-1]]) < 1e-12
    assert np.all(u[:, ~fl] == 0.0)
    lb.free()


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("mode", [0, 1, 2, 3], ids=["eager", "fused", "inplace", "fused_halo"])
@pytest.mark.parametrize("nvel", [19, 27])
def test_hydro_arrays_with_their_own_component_stride(nvel, mode, lazy):
    """lbmi_hydro_t::nsite: force and u whose components lie further apart
    than the lattice's nsite (with Lees-Edwards planes the reference allocates
    the hydro arrays with buffer planes, hydro.c:75-88). Same results as with
    the plain arrays; the padding between the components is never touched."""
    import ludwig_amd
    import torch
    nsteps = 3
    nall = tuple(n + 2 for n in NLOCAL)
    nsite = int(np.prod(nall))
    stride = nsite + 4 * nall[1] * nall[2] + 3          # buffer planes and an odd tail
    force = _force_field(nall)
    status = _status(nall)
    p, f_o, rho_o, u_o = _oracle_run(nvel, nsteps, force, status)

    lb = ludwig_amd.LB(nvel, NLOCAL, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    hy = ludwig_amd.Hydro(nall, lb.device, status=status)
    marker = -7.25
    fpad = np.full((3, stride), marker)
    fpad[:, :nsite] = force.reshape(3, nsite)
    hy.force = torch.from_numpy(fpad).to(lb.device)
    hy.u = torch.full((3, stride), marker, dtype=torch.float64, device=lb.device)
    hy.stride = stride
    torch.cuda.synchronize(lb.device)
    if lazy and mode in (1, 3):
        lb.tune("hydro_lazy", 1)
    lb.lb_memcpy_h2d(lbo.init_synthetic(p))
    for _ in range(nsteps):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.lb_propagation()
    lb.hydro_sync()
    f = lb.lb_memcpy_d2h()
    lb.synchronize()
    assert relmax(interior(f, 1), interior(f_o, 1)) < 1e-12
    fl = _fluid(status, nall)
    u = hy.u.cpu().numpy()
    ug = u[:, :nsite].reshape((3,) + nall)
    assert relmax(interior(ug, 1)[:, fl], interior(u_o, 1)[:, fl]) < 1e-12
    assert relmax(interior(hy.rho.cpu().numpy(), 1)[fl], interior(rho_o, 1)[fl]) < 1e-12
    # nothing between the components has been written, nor read as a force
    assert np.all(u[:, nsite:] == marker)
    lb.free()


@pytest.mark.parametrize("mode", [1, 3], ids=["fused", "fused_halo"])
def test_rho_alone_on_demand(mode):
    """hydro_lazy 2: u stored by every collision (equal to the oracle's after
    every step, no sync), rho formed when asked for -- with a force field, solid
    sites, at several call points; the distributions are untouched by it."""
    import ludwig_amd
    import torch
    nall = tuple(n + 2 for n in NLOCAL)
    force = _force_field(nall)
    status = np.zeros(nall, dtype=np.int8)
    status[3:5, 2:4, 4:7] = 1
    nsteps = 4
    p = lbo.make_param(19, NLOCAL, 1, "m10", 0.1, 0.3, 1.0, FBODY)
    f = lbo.init_synthetic(p)
    f0 = f.copy()
    fp = np.zeros_like(f)
    rho = np.zeros(nall)
    u = np.zeros((3,) + nall)
    lb = ludwig_amd.LB(19, NLOCAL, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set(FBODY)
    lb.tune("hydro_lazy", 2)
    hy = ludwig_amd.Hydro(nall, lb.device, force=force, status=status)
    lb.lb_memcpy_h2d(f0)
    fluid = (status == 0)[1:-1, 1:-1, 1:-1]
    for n in range(nsteps):
        lbo.collide(p, f, force, status, rho, u)
        lb.lb_collide(hy)
        lb.synchronize()
        torch.cuda.synchronize()
        assert relmax(interior(hy.u.cpu().numpy(), 1)[:, fluid], interior(u, 1)[:, fluid]) < 1e-12
        if n == 1:
            lb.hydro_sync()                       # between lb_collide and lb_halo
        lbo.halo(p, f)
        lb.lb_halo()
        lbo.propagate(p, f, fp)
        f, fp = fp, f
        lb.lb_propagation()
        if n in (1, 2):
            lb.hydro_sync()                       # with the propagation pending
            lb.synchronize()
            torch.cuda.synchronize()
            assert relmax(interior(hy.rho.cpu().numpy(), 1)[fluid], interior(rho, 1)[fluid]) < 1e-12
    out = lb.lb_memcpy_d2h()                      # the flush settles what is owed
    assert relmax(interior(hy.rho.cpu().numpy(), 1)[fluid], interior(rho, 1)[fluid]) < 1e-12
    assert relmax(interior(out, 1)[:, fluid], interior(f, 1)[:, fluid]) < 1e-12
    lb.free()
