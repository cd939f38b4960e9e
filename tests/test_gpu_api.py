"""GPU tests of the remaining C-ABI surface: tuning knobs do not change
results, error paths, moments on caller-owned arrays, timing counters."""

import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import lb_oracle as lbo                       # noqa: E402
from tests.common import interior, relmax                  # noqa: E402


def _run(tune, mode, nvel=19, nlocal=(20, 9, 14), nsteps=5):
    import ludwig_amd
    p = lbo.make_param(nvel, nlocal, 1, "m10", 0.1, 0.3, 1.0, (1e-6, 0, 0))
    f0 = lbo.init_synthetic(p)
    lb = ludwig_amd.LB(nvel, nlocal, 1, mode=mode)
    lb.relaxation_set("m10", 0.1, 0.3)
    lb.body_force_set((1e-6, 0, 0))
    for k, v in tune.items():
        lb.tune(k, v)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.lb_memcpy_h2d(f0)
    for _ in range(nsteps):
        lb.step(hy)
    out = interior(lb.lb_memcpy_d2h(), 1).copy()
    lb.synchronize()
    u = interior(hy.u.cpu().numpy(), 1).copy()
    lb.free()
    return out, u


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["eager", "fused", "inplace"])
def test_tuning_does_not_change_results(mode):
    base, ubase = _run({"xcd_group": 0, "lds_cap": 0}, mode)
    for tune in ({"xcd_group": 16, "lds_cap": 65536},
                 {"xcd_group": 3, "lds_cap": 40000},
                 {"xcd_group": 1, "lds_cap": 98304}):
        out, u = _run(tune, mode)
        assert np.array_equal(out, base)
        assert np.array_equal(u, ubase)


def test_tune_rejects_unknown_key():
    import ludwig_amd
    lb = ludwig_amd.LB(19, (4, 4, 4))
    with pytest.raises(ludwig_amd.LbmiError):
        lb.tune("no_such_knob", 1)
    with pytest.raises(ludwig_amd.LbmiError):
        lb.tune("lds_cap", 1 << 20)
    lb.free()


def test_moments_match_oracle_with_status():
    import ludwig_amd
    import torch
    nlocal = (12, 10, 8)
    p = lbo.make_param(27, nlocal, 2, "m10")
    f = lbo.init_synthetic(p)
    st = np.zeros(lbo.nall(p), dtype=np.int8)
    st[3:6, 4:7, 2:9] = 1
    ref = lbo.moments(p, f, st)
    lb = ludwig_amd.LB(27, nlocal, 2)
    t = torch.from_numpy(f).to(lb.device)
    s = torch.from_numpy(st).to(lb.device)
    torch.cuda.synchronize()
    mo = lb.moments_of(t, s)
    assert mo[0] == ref[0]
    assert abs(mo[1] - ref[1]) / ref[1] < 1e-14
    assert abs(mo[2] - ref[2]) / ref[2] < 1e-14
    assert mo[3] == ref[3] and mo[4] == ref[4]        # min / max exact
    assert np.max(np.abs(mo[5:8] - ref[5:8])) < 1e-13
    # bitwise reproducible run to run (no atomics)
    assert np.array_equal(mo, lb.moments_of(t, s))
    lb.free()


def test_timing_counts_fused_launches():
    import ludwig_amd
    lb = ludwig_amd.LB(19, (16, 16, 16), 1, mode=ludwig_amd.FUSED)
    lb.lb_memcpy_h2d(lbo.init_synthetic(lbo.make_param(19, (16, 16, 16))))
    lb.step()
    lb.timing(True)
    for _ in range(7):
        lb.step()
    ms, n = lb.timing_read()
    assert n == 7 and ms > 0.0
    lb.free()


def test_unbound_handle_is_a_state_error():
    import ctypes
    from ludwig_amd import lib as L
    lib = L.library()
    o = L.Options()
    h = ctypes.c_void_p()
    lib.lbmi_options_default(ctypes.byref(o))
    o.nlocal[:] = [4, 4, 4]
    assert lib.lbmi_create(ctypes.byref(o), ctypes.byref(h)) == 0
    assert lib.lbmi_lb_collide(h, None) == -6          # LBMI_ERR_STATE
    assert b"no distributions bound" in lib.lbmi_last_error()
    assert lib.lbmi_free(h) == 0


# --- rows "next": hydro housekeeping and the distribution record stream -------

@pytest.mark.parametrize("mode", [0, 1, 2], ids=["eager", "fused", "inplace"])
@pytest.mark.parametrize("name", ["q19_m10", "q19_m10_nh2_ffield", "q27_m10"])
def test_record_stream_matches_reference(name, mode):
    """lb_io_aggr_pack / lb_write_buf (model.c:1385-1510): the records of the
    state after nsteps equal the reference's, whatever the execution mode."""
    import ludwig_amd
    from tests.common import RTOL_F, load_golden
    g = load_golden(name)
    meta = g["meta"]
    lb = ludwig_amd.LB(meta["nvel"], tuple(meta["nlocal"]), meta["nhalo"], mode=mode)
    lb.relaxation_set(meta["scheme"], meta["eta"], meta["zeta"])
    lb.body_force_set(meta["fbody"])
    hy = ludwig_amd.Hydro(lb.nall, lb.device, force=g["force"])
    lb.lb_memcpy_h2d(g["f0"])
    for _ in range(meta["nsteps"]):
        lb.step(hy)
    rec = lb.lb_io_aggr_pack()
    assert rec.shape == g["records"].shape
    assert relmax(rec, g["records"]) < RTOL_F
    # the stream is a pure re-ordering of f: exact against the D2H copy
    h = meta["nhalo"]
    f = interior(lb.lb_memcpy_d2h(), h)
    assert np.array_equal(np.moveaxis(rec, 3, 0), f)
    lb.free()


@pytest.mark.parametrize("nvel,nlocal,nhalo", [(19, (7, 5, 3), 1), (27, (4, 9, 70), 2),
                                                (19, (1, 1, 300), 1)])
def test_record_stream_round_trip(nvel, nlocal, nhalo):
    """unpack(pack(f)) == f on the interior, bit for bit; halo untouched;
    and the device kernels agree with the oracle's restatement."""
    import ludwig_amd
    p = lbo.make_param(nvel, nlocal, nhalo)
    rng = np.random.default_rng(11)
    f = rng.standard_normal((nvel,) + lbo.nall(p))
    lb = ludwig_amd.LB(nvel, nlocal, nhalo)
    lb.lb_memcpy_h2d(f)
    rec = lb.lb_io_aggr_pack()
    assert np.array_equal(rec, lbo.records_pack(p, f))
    lb.lb_memcpy_h2d(np.full_like(f, -1.0))
    lb.lb_io_aggr_unpack(rec)
    out = lb.lb_memcpy_d2h()
    assert np.array_equal(interior(out, nhalo), interior(f, nhalo))
    halo_mask = np.ones(out.shape, dtype=bool)
    interior(halo_mask, nhalo)[...] = False
    assert np.all(out[halo_mask] == -1.0)
    lb.free()


def test_hydro_field_set():
    # hydro_u_zero / hydro_f_zero: all sites, halo included (hydro.c:279-330)
    import ludwig_amd
    import torch
    lb = ludwig_amd.LB(19, (5, 4, 3), 2)
    u = torch.full((3,) + lb.nall, 9.0, dtype=torch.float64, device=lb.device)
    rho = torch.full(lb.nall, 9.0, dtype=torch.float64, device=lb.device)
    torch.cuda.synchronize()
    lb.hydro_field_set(u, (0.5, -1.5, 2.5))
    lb.hydro_field_set(rho, (1.0,))
    lb.synchronize()
    un = u.cpu().numpy()
    assert np.all(un[0] == 0.5) and np.all(un[1] == -1.5) and np.all(un[2] == 2.5)
    assert np.all(rho.cpu().numpy() == 1.0)
    lb.free()


def test_lb_run_equals_steps():
    """lbmi_lb_run(n) = n x (collide, halo, propagation), every mode."""
    import ludwig_amd
    from oracle import lb_oracle as lbo
    nlocal = (20, 12, 16)
    p = lbo.make_param(19, nlocal, 1, "m10", 0.1, 0.3)
    f0 = lbo.init_synthetic(p)
    for mode in (ludwig_amd.EAGER, ludwig_amd.FUSED, ludwig_amd.INPLACE):
        out = []
        for use_run in (False, True):
            lb = ludwig_amd.LB(19, nlocal, 1, mode=mode)
            lb.relaxation_set("m10", 0.1, 0.3)
            hy = ludwig_amd.Hydro(lb.nall, lb.device)
            lb.lb_memcpy_h2d(f0)
            if use_run:
                lb.run(hy, 5)
            else:
                for _ in range(5):
                    lb.step(hy)
            out.append(lb.lb_memcpy_d2h()[:, 1:-1, 1:-1, 1:-1].copy())
            lb.free()
        assert np.array_equal(out[0], out[1])


def test_nt_store_auto_and_explicit_agree():
    import ludwig_amd
    from oracle import lb_oracle as lbo
    nlocal = (24, 10, 18)
    p = lbo.make_param(19, nlocal, 1, "bgk", 0.1, 0.1)
    f0 = lbo.init_synthetic(p)
    out = []
    for nt in (-1, 0, 1, 3):
        lb = ludwig_amd.LB(19, nlocal, 1, mode=ludwig_amd.FUSED)
        lb.tune("nt_store", nt)
        lb.relaxation_set("bgk", 0.1, 0.1)
        hy = ludwig_amd.Hydro(lb.nall, lb.device)
        lb.lb_memcpy_h2d(f0)
        lb.run(hy, 4)
        out.append((lb.lb_memcpy_d2h()[:, 1:-1, 1:-1, 1:-1].copy(),
                    hy.u.cpu().numpy()))
        lb.free()
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])


@pytest.mark.parametrize("mode", [1, 3], ids=["fused", "fused_halo"])
@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_self_ring(mode, seed, tmp_path):
    """The same fuzz on the slab path: the X halo through a 1-rank RCCL
    ring (interior / exchange / boundary planes, blocked order where the
    lattice allows it)."""
    _fuzz(mode, 500 + seed, tmp_path, ring=True)


@pytest.mark.parametrize("mode", [1, 2, 3], ids=["fused", "inplace", "fused_halo"])
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("LBMI_FUZZ_SEEDS", "12"))))
def test_random_call_sequences_match_eager(mode, seed, tmp_path):
    """State-machine fuzz: a random but legal interleaving of the step calls
    with observers (device-to-host copies, moments, record packs, file
    round trips, lbmi_lb_run) and tuning switches; every observation must be
    what EAGER shows at the same point of the same sequence. Since round 2
    the sequences also switch rho, u on demand on and off, ask for u
    (lbmi_lb_hydro_sync) at any point, and change the execution mode of the
    handle in mid-run (lbmi_lb_mode_set)."""
    _fuzz(mode, seed, tmp_path, ring=False)


def _fuzz(mode, seed, tmp_path, ring):
    import ludwig_amd
    rng = np.random.default_rng(1000 + seed)
    nvel = 19 if seed % 3 else 27
    nlocal = (10, 14, 14) if seed % 2 else (9, 7, 12)     # whole blocks / ragged
    p = lbo.make_param(nvel, nlocal, 1, "m10", 0.1, 0.3, 1.0, (1e-6, 0, -2e-6))
    f0 = lbo.init_synthetic(p)
    nsteps = 6
    # some sequences with a force field, a block of solid sites, a local
    # viscosity field (the optional inputs of lb_collide)
    nall = lbo.nall(p)
    force = 1e-6 * rng.standard_normal((3,) + nall) if seed % 4 == 1 else None
    status = None
    if seed % 5 == 2:
        status = np.zeros(nall, dtype=np.int8)
        status[2:4, 2:4, 2:5] = 1
    eta = 0.1 * (1.0 + 0.3 * rng.random(nall)) if seed % 7 == 3 else None
    # the script: one list of actions, generated once, run in both modes
    script = []
    for n in range(nsteps):
        for stage in ("collide", "halo", "propagation"):
            script.append((stage,))
            r = rng.random()
            if r < 0.15:
                script.append(("d2h",))
            elif r < 0.25:
                script.append(("moments",))
            elif r < 0.32:
                script.append(("records",))
            elif r < 0.40:
                script.append(("u",))
            elif r < 0.45 and stage == "collide":
                # a writer of f between lb_collide and lb_halo that goes through
                # a flush and owns up (the reference's lb_memcpy both ways)
                script.append(("dirty",))
        r2 = rng.random()
        if r2 < 0.2:
            script.append(("tune", "hydro_lazy", int(rng.integers(0, 2))))
        elif r2 < 0.3:
            script.append(("mode_set", int(rng.choice([0, 1, 3]))))
        elif r2 < 0.4 and ring:
            script.append(("tune", "x_direct", int(rng.integers(0, 2))))
        r = rng.random()
        if r < 0.2:
            script.append(("tune", "blocked", int(rng.integers(0, 2))))
        elif r < 0.35:
            script.append(("tune", "nt_store", int(rng.integers(-1, 4))))
        elif r < 0.45:
            script.append(("tune", "xcd_group", int(rng.choice([0, 1, 4, 32]))))
        elif r < 0.5 and ring:
            script.append(("tune", "x_concurrent", int(rng.integers(0, 2))))
        elif r < 0.55:
            script.append(("io", n))
        elif r < 0.7:
            script.append(("run", int(rng.integers(1, 4))))
        elif r < 0.8:
            script.append(("tune", "halo_fold", int(rng.integers(0, 2))))

    def play(run_mode):
        # (the self-ring along X, Y or Z: slabs of any direction run the same
        # scripts; Y and Z: gathered planes, the face launch of the fused step)
        lb = ludwig_amd.LB(nvel, nlocal, 1, mode=run_mode,
                           halo_scheme=2 if ring else 0,
                           cartdim=(seed % 3) if ring else 0)
        if ring:
            lb.comm_init(ludwig_amd.LB.comm_unique_id())
        lb.relaxation_set("m10", 0.1, 0.3)
        lb.body_force_set((1e-6, 0, -2e-6))
        hy = ludwig_amd.Hydro(lb.nall, lb.device, force=force, status=status,
                              eta=eta)
        lb.lb_memcpy_h2d(f0)
        seen = []
        for act in script:
            if act[0] == "collide":
                lb.lb_collide(hy)
            elif act[0] == "halo":
                lb.lb_halo()
            elif act[0] == "propagation":
                lb.lb_propagation()
            elif act[0] == "d2h":
                seen.append(interior(lb.lb_memcpy_d2h(), 1).copy())
            elif act[0] == "moments":
                seen.append(lb.moments(hy.status))
            elif act[0] == "records":
                seen.append(lb.lb_io_aggr_pack().copy())
            elif act[0] == "u":
                # rho, u of the last collision: owed by a lazy one, there already otherwise
                lb.hydro_sync()
                seen.append(host_u(lb, hy))
            elif act[0] == "mode_set":
                if run_mode != 0:
                    lb.mode_set(act[1])
            elif act[0] == "tune":
                lb.tune(act[1], act[2])
            elif act[0] == "io":
                lb.lb_io_write(tmp_path, act[1])
                lb.lb_io_read(tmp_path, act[1])
            elif act[0] == "run":
                lb.run(hy, act[1])
            elif act[0] == "dirty":
                import torch
                lb.lb_flush()
                lb.synchronize()
                lb.f[:, 2, :, :] *= 1.0 + 1e-3
                torch.cuda.synchronize()
                lb.lb_dirty()
        seen.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.hydro_sync()
        seen.append(host_u(lb, hy))
        lb.free()
        return seen

    def host_u(lb, hy):
        lb.synchronize()
        return interior(hy.u.cpu().numpy(), 1).copy()

    ref = play(0)
    out = play(mode)
    assert len(ref) == len(out)
    for k, (a, b) in enumerate(zip(ref, out)):
        scale = max(1.0, float(np.max(np.abs(a))))
        assert np.max(np.abs(np.asarray(a) - np.asarray(b))) < 1e-13 * scale, (k, script)


@pytest.mark.parametrize("own_stream", [False, True], ids=["torch_stream", "own_stream"])
@pytest.mark.parametrize("nvel,scheme", [(19, "m10"), (19, "trt"), (27, "bgk")])
def test_lb_run_as_graph_equals_steps(nvel, scheme, own_stream):
    """tune("graph", 1): lbmi_lb_run launches pairs of steady-state steps as
    one hipGraph. Same results as step by step, bit for bit: odd and even
    counts, from a fresh and from a flushed handle, after the parameters or the
    arrays changed (the graph is keyed on them), and observers in between."""
    import ludwig_amd
    import torch
    n = (24, 12, 20)
    p = lbo.make_param(nvel, n, 1, scheme, 0.1, 0.2)
    f0 = lbo.init_synthetic(p)
    frc = 1e-6 * np.random.default_rng(3).standard_normal((3,) + lbo.nall(p))
    out = []
    for graph in (0, 1):
        lb = ludwig_amd.LB(nvel, n, 1, mode=ludwig_amd.FUSED, own_stream=own_stream)
        lb.relaxation_set(scheme, 0.1, 0.2)
        lb.tune("graph", graph)
        hy = ludwig_amd.Hydro(lb.nall, lb.device, force=frc)
        torch.cuda.synchronize()
        lb.lb_memcpy_h2d(f0)
        rec = []
        lb.run(hy, 9)                                   # odd: 2 + 3 pairs + 1
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())      # flushes
        lb.run(hy, 12)                                  # from a flushed handle
        lb.run(hy, 7)                                   # graph re-used or re-keyed
        rec.append(lb.moments())
        lb.body_force_set((1e-6, 0, -2e-6))             # parameters by value in the graph
        lb.run(hy, 8)
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.step(hy)                                     # parity of f / fprime changes
        lb.run(hy, 10)
        lb.synchronize()
        rec.append(interior(hy.u.cpu().numpy(), 1).copy())
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.run(hy, 3)                                   # too short for the graph
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.free()
        out.append(rec)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_a_captured_run_does_not_outlive_what_its_launches_carried():
    """lbmi_lb_collide joins state of the handle to the hydro object at the
    moment of the call -- whether the force array is known to hold zeros,
    whether rho and u are stored, the generator and temperature of the
    fluctuations -- and a captured lbmi_lb_run (tune "graph") holds the
    launches as they were captured. Each change between two runs must
    re-capture: a force field zeroed through the library and then written
    (reported dirty), the lazy switch flipped both ways, fluctuations switched
    on, another temperature. Bit for bit against the same calls without the
    graph."""
    import ludwig_amd
    import torch
    n = (24, 12, 20)
    p = lbo.make_param(19, n, 1, "m10", 0.1, 0.2)
    f0 = lbo.init_synthetic(p)
    nall = lbo.nall(p)
    frc = 1e-6 * np.random.default_rng(5).standard_normal((3,) + nall)
    nsite = int(np.prod(nall))
    state0 = np.random.default_rng(6).integers(1, 2**31 - 1, size=(4, nsite)).astype(np.uint32)
    out = []
    for graph in (0, 1):
        lb = ludwig_amd.LB(19, n, 1, mode=ludwig_amd.FUSED)
        lb.relaxation_set("m10", 0.1, 0.2)
        lb.tune("graph", graph)
        hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros_like(frc))
        state = torch.from_numpy(state0.view(np.int32).copy()).to(lb.device)
        torch.cuda.synchronize()
        lb.lb_memcpy_h2d(f0)
        rec = []
        lb.hydro_field_set(hy.force, (0.0, 0.0, 0.0))   # known zero: not read
        lb.run(hy, 8)
        hy.force.copy_(torch.from_numpy(frc))           # written (the mirror reports it)
        torch.cuda.synchronize()
        lb.run(hy, 8)                                   # the captured launches skipped it
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        lb.tune("hydro_lazy", 1)
        lb.run(hy, 8)
        lb.tune("hydro_lazy", 0)                        # stored again from here
        hy.u.zero_()
        torch.cuda.synchronize()
        lb.run(hy, 8)
        lb.synchronize()
        rec.append(interior(hy.u.cpu().numpy(), 1).copy())
        lb.noise_set(state, 1e-5)                       # fluctuations on
        lb.run(hy, 8)
        lb.noise_set(state, 4e-5)                       # another temperature
        lb.run(hy, 8)
        lb.noise_set(None, 0.0)
        lb.run(hy, 8)
        rec.append(interior(lb.lb_memcpy_d2h(), 1).copy())
        rec.append(state.cpu().numpy().copy())
        lb.free()
        out.append(rec)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    # and the changes did change something (the test would pass trivially)
    assert np.abs(out[0][1]).max() > 0.0
    assert not np.array_equal(out[0][3], state0.view(np.int32))


@pytest.mark.parametrize("nlocal,nhalo", [((9, 7, 12), 1), ((16, 14, 14), 2), ((1, 5, 3), 1)])
def test_field_stats_match_numpy(nlocal, nhalo):
    """lbmi_field_stats (cahn_stats_reduce, cahn_hilliard_stats.c:123-215):
    volume, compensated sum, sum of squares, extrema of a scalar field over
    the interior sites that are fluid; halo and solid sites hold values that
    would show if they were counted."""
    import ludwig_amd
    import torch
    lb = ludwig_amd.LB(19, nlocal, nhalo)
    rng = np.random.default_rng(3)
    phi = 1e6 * np.ones(lb.nall)                          # halo: poison
    interior(phi, nhalo)[...] = 0.3 * rng.standard_normal(nlocal) + 1e-3
    status = np.zeros(lb.nall, dtype=np.int8)
    h = nhalo
    if nlocal[0] > 4:
        status[h + 1:h + 3, h:h + 2, h + 1:h + 4] = 1
        phi[h + 1:h + 3, h:h + 2, h + 1:h + 4] = -1e6    # solid: poison
    dphi = torch.from_numpy(phi).to(lb.device)
    dst = torch.from_numpy(status).to(lb.device)
    torch.cuda.synchronize()
    for st, mask in ((dst, interior(status, h) == 0), (None, None)):
        if st is None:
            if nlocal[0] > 4:
                continue                                   # (the poison would count)
            vals = interior(phi, h).ravel()
        else:
            vals = interior(phi, h)[mask]
        out = lb.field_stats(dphi, st)
        assert out[0] == vals.size
        assert abs(out[1] - math.fsum(vals)) <= 4e-16 * np.abs(vals).sum()
        assert abs(out[2] - math.fsum(vals * vals)) <= 1e-14 * (vals * vals).sum()
        assert out[3] == vals.min() and out[4] == vals.max()
    lb.free()
