"""Isothermal fluctuations in lb_collide (lbmi_noise_set; collision.c:476-518,
noise.c:397-487) on the MI355X against the compiled reference: fixtures
noise_q19_* (M10, TRT with a solid block and a force field, BGK with a
viscosity model, M10 with ghost modes off). Distributions to 1e-12; the
generator states -- integer arithmetic -- bit for bit. EAGER (collision in
place), FUSED_HALO (the binding's default: the pull of the previous step's
propagation folded into the fluctuating collision), FUSED (the halo swap
folded in as well) and the slab step of FUSED on a one-rank ring."""

import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                                   # noqa: E402
from tests.common import (GOLDEN_DIR, RTOL_F, interior, load_golden,   # noqa: E402
                          relmax, status_from_meta)


def noise_names():
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "noise_q19_*.npz")))


def _setup(g, mode):
    import torch
    import ludwig_amd
    meta = g["meta"]
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), meta["nhalo"], mode=mode)
    lb.relaxation_set(meta["scheme"], meta["eta"], meta["zeta"], meta["rho0"])
    lb.body_force_set(meta["fbody"])
    st = status_from_meta(meta)
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=g["force"],
                          status=st if meta["solid"] else None,
                          eta=g["eta"] if meta["visc"] else None)
    nsite = int(np.prod(lb.nall))
    state = torch.from_numpy(g["noise0"].reshape(4, nsite).view(np.int32).copy()).to(lb.device)
    torch.cuda.synchronize(lb.device)
    lb.noise_set(state, meta["kt"], meta["ghosts"])
    lb.lb_memcpy_h2d(g["f0"])
    return lb, hy, state, st


def _state_host(lb, state, nall):
    lb.synchronize()
    return state.cpu().numpy().view(np.uint32).reshape((4,) + tuple(nall))


@pytest.mark.parametrize("name", noise_names())
def test_first_collision_vs_reference(name):
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb, hy, state, st = _setup(g, ludwig_amd.EAGER)
    lb.lb_collide(hy)
    f = lb.lb_memcpy_d2h()
    fluid = interior(st, h) == 0
    assert relmax(interior(f, h), interior(g["f_collide"], h)) < RTOL_F
    lb.synchronize()
    assert relmax(interior(hy.rho.cpu().numpy(), h)[fluid], interior(g["rho"], h)[fluid]) < RTOL_F
    assert relmax(interior(hy.u.cpu().numpy(), h)[:, fluid], interior(g["u"], h)[:, fluid]) < RTOL_F
    s1 = _state_host(lb, state, lb.nall)
    if meta["solid"]:
        # solid sites neither collide nor draw
        assert np.array_equal(interior(f, h)[:, ~fluid], interior(g["f0"], h)[:, ~fluid])
        assert np.array_equal(interior(s1, h)[:, ~fluid], interior(g["noise0"], h)[:, ~fluid])
    # every fluid site has drawn: the oracle's states after one collision
    p = lbo.make_param(meta["nvel"], meta["nlocal"], h, meta["scheme"], meta["eta"],
                       meta["zeta"], meta["rho0"], meta["fbody"])
    so = g["noise0"].copy()
    lbo.collide_noise(p, g["f0"].copy(), g["force"].copy(), st, so, meta["kt"], meta["ghosts"],
                      eta=g["eta"].copy() if meta["visc"] else None)
    assert np.array_equal(interior(s1, h), interior(so, h))
    lb.free()


@pytest.mark.parametrize("mode", ["eager", "fused_halo", "fused", "fused_ring"])
@pytest.mark.parametrize("name", noise_names())
def test_steps_and_generator_states_vs_reference(name, mode):
    """fused: halo swap and propagation folded into the fluctuating collision
    (index wrap; the deferred state in the blocked order where the lattice
    allows it); fused_ring: the same through a one-rank RCCL ring, i.e. the
    slab step (interior and boundary launches, pack / messages / unpack)."""
    import ludwig_amd
    g = load_golden(name)
    meta = g["meta"]
    h = meta["nhalo"]
    lb, hy, state, st = _setup(g, {"eager": ludwig_amd.EAGER, "fused_halo": ludwig_amd.FUSED_HALO,
                                   "fused": ludwig_amd.FUSED, "fused_ring": ludwig_amd.FUSED}[mode])
    if mode == "fused_ring":
        lb.comm_init(ludwig_amd.LB.comm_unique_id())
    for _ in range(meta["nsteps"]):
        lb.lb_collide(hy)
        lb.lb_halo()
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, h), interior(g["f_final"], h)) < RTOL_F
    assert np.array_equal(interior(_state_host(lb, state, lb.nall), h),
                          interior(g["noise_final"], h))
    lb.free()


def test_the_c_loop_of_the_library_with_fluctuations():
    """lbmi_lb_run (the three calls of every step issued from C): the same
    states and distributions as the calls one by one."""
    import ludwig_amd
    g = load_golden("noise_q19_trt_solid")
    meta = g["meta"]
    h = meta["nhalo"]
    lb, hy, state, st = _setup(g, ludwig_amd.FUSED_HALO)
    lb.run(hy, meta["nsteps"])
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, h), interior(g["f_final"], h)) < RTOL_F
    assert np.array_equal(interior(_state_host(lb, state, lb.nall), h),
                          interior(g["noise_final"], h))
    lb.free()


def test_switching_off_and_on_again():
    """lbmi_noise_set(NULL): the plain collision again, generator untouched."""
    import ludwig_amd
    g = load_golden("noise_q19_m10")
    meta = g["meta"]
    h = meta["nhalo"]
    lb, hy, state, st = _setup(g, ludwig_amd.EAGER)
    lb.noise_set(None, 0.0)
    lb.lb_collide(hy)
    f = lb.lb_memcpy_d2h()
    p = lbo.make_param(meta["nvel"], meta["nlocal"], h, meta["scheme"], meta["eta"],
                       meta["zeta"], meta["rho0"], meta["fbody"])
    fo = g["f0"].copy()
    lbo.collide(p, fo, g["force"].copy(), st)
    assert relmax(interior(f, h), interior(fo, h)) < RTOL_F
    assert np.array_equal(_state_host(lb, state, lb.nall), g["noise0"])
    lb.free()


def test_what_fluctuations_do_not_cover_is_refused():
    import torch
    import ludwig_amd
    # D3Q27: the reference's generator cannot serve its 17 ghost modes (noise.h:18)
    lb = ludwig_amd.LB(27, (4, 4, 4), 1)
    state = torch.ones((4, int(np.prod(lb.nall))), dtype=torch.int32, device=lb.device)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.noise_set(state, 1e-4)
    lb.free()
    # too few generator states
    lb = ludwig_amd.LB(19, (4, 4, 4), 1)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.noise_set(state[:, :10].contiguous(), 1e-4)
    lb.free()
    # INPLACE (the AA pair): lb_collide says so
    for mode in (ludwig_amd.INPLACE,):
        lb = ludwig_amd.LB(19, (4, 4, 4), 1, mode=mode)
        lb.relaxation_set("m10", 0.1, 0.3)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        st19 = torch.ones((4, int(np.prod(lb.nall))), dtype=torch.int32, device=lb.device)
        lb.noise_set(st19, 1e-4)
        lb.lb_memcpy_h2d(lbo.init_synthetic(lbo.make_param(19, (4, 4, 4), 1, "m10", 0.1, 0.3, 1.0, (0, 0, 0))))
        with pytest.raises(ludwig_amd.LbmiError):
            lb.lb_collide(hy)
        lb.free()


# --- the two-distribution collision with fluctuations (collision.c:884-900) ---

@pytest.mark.parametrize("mode", [0, 3, 1], ids=["eager", "fused_halo", "fused"])
def test_binary_steps_with_fluctuations_vs_reference(mode):
    """free_energy symmetric_lb + isothermal_fluctuations: lb_collision_binary
    draws at every site (no status test); every execution mode."""
    import torch
    import ludwig_amd
    g = load_golden("noise_bin_q19_a")
    meta = g["meta"]
    lb = ludwig_amd.LB(19, tuple(meta["nlocal"]), 1, ndist=2, mode=mode)
    lb.relaxation_set("m10", meta["eta"], meta["zeta"])
    lb.body_force_set(meta["fbody"])
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.fe_scheme_set(27, 1)
    nsite = int(np.prod(lb.nall))
    state = torch.from_numpy(g["noise0"].reshape(4, nsite).view(np.int32).copy()).to(lb.device)
    phi = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    grad = torch.zeros((3,) + lb.nall, dtype=torch.float64, device=lb.device)
    delsq = torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize(lb.device)
    lb.noise_set(state, meta["kt"], True)
    lb.lb_memcpy_h2d(g["f0"])
    for n in range(meta["nsteps"]):
        lb.phi_to_field(phi)
        lb.field_halo_n(phi, 1)
        lb.field_grad(phi, grad, delsq)
        lb.hydro_field_set(hy.u, (0, 0, 0))
        lb.lb_collide_binary(hy, meta["a"], meta["b"], meta["kappa"], meta["mobility"],
                             phi, grad, delsq)
        if n == 0 and mode == 0:
            f = lb.lb_memcpy_d2h()
            assert relmax(interior(f, 1), interior(g["f_collide"], 1)) < RTOL_F
        lb.lb_halo()
        lb.lb_propagation()
    f = lb.lb_memcpy_d2h()
    assert relmax(interior(f, 1), interior(g["f_final"], 1)) < RTOL_F
    assert np.array_equal(interior(_state_host(lb, state, lb.nall), 1),
                          interior(g["noise_final"], 1))
    lb.free()
