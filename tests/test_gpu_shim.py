"""The drop-in boundary, end to end: the reference's OWN HIP target
(target/target_hip.c and its TargetDP kernels, compiled for gfx950 by
oracle/Makefile target "hip" from the sources where they lie) with
integration/ludwig_shim.c bound in by the renames of INTEGRATION.md and linked
against liblbmi.so. The same driver that produced the golden fixtures on the
CPU calls lb_collide / lb_halo / lb_propagation / lb_memcpy of that build; its
dumps must be the fixtures. Needs an MI355X and the prebuilt binaries in
oracle/_ref/ (they travel with the snapshot; /root/reference does not)."""

import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import make_golden as mg                          # noqa: E402
from tests.common import interior, load_golden, relmax       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref")
KEYS = ("f_collide", "rho", "u", "f_halo", "f_prop", "f_final")


def _exe(nvel, shim):
    exe = os.path.join(REF, "ref_driver_hip_d3q%d%s" % (nvel, "_shim" if shim else ""))
    if not os.path.exists(exe):
        # (not a skip: the binaries travel with the snapshot, and a missing one
        # would silently take the whole boundary out of the suite)
        pytest.fail("oracle/_ref/%s is missing: build it in the development "
                    "container with `make -C oracle hip`" % os.path.basename(exe))
    return exe


def _compare(out, g, nhalo):
    for key in KEYS:
        if key not in g or key not in out:
            continue
        a, b = out[key], g[key]
        if key in ("rho", "u", "f_collide", "f_prop", "f_final"):
            a, b = interior(a, nhalo), interior(b, nhalo)
        elif key == "f_halo":
            # the width-1 shell around the interior is what lb_halo fills
            h = nhalo - 1
            if h:
                a, b = a[..., h:-h, h:-h, h:-h], b[..., h:-h, h:-h, h:-h]
        assert relmax(a, b) < 1e-12, key


M10_CASES = [c for c in mg.CASES if c[4] == "m10"]


def _env(mode, **extra):
    """The child's environment: LBMI_MODE = mode, or unset for None (the
    binding chooses: fused until something needs the reference's state between
    lb_collide and lb_propagation); no other LBMI_* switch unless given."""
    env = dict(os.environ)
    for k in ("LBMI_MODE", "LBMI_FE", "LBMI_HYDRO", "LBMI_REPORT"):
        env.pop(k, None)
    if mode is not None:
        env["LBMI_MODE"] = mode
    env.update(extra)
    return env


@pytest.mark.parametrize("case", M10_CASES, ids=[c[0] for c in M10_CASES])
def test_reference_hip_target_reproduces_its_cpu_fixtures(case):
    """No binding: the reference's TargetDP kernels on the MI355X against the
    fixtures its CPU build produced (validates the HIP build of the driver).
    M10 only: on a device the reference's kernels read lb->nrelax from the
    DEVICE copy of lb_t, which nothing ever sets (model.c:81 and collision.c:
    1161 write the host struct only), so its GPU builds relax the ghost and
    bulk modes as M10 whatever scheme was asked for -- its BGK and TRT results
    differ from its own CPU path at the 1e-3 level. The parity target is the
    CPU path (BASELINE north_star); the binding reads the host struct."""
    exe = _exe(case[1], shim=False)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_case(case, tmp, exe=exe)
    _compare(out, load_golden(case[0]), case[3])


@pytest.mark.parametrize("mode", ["eager", "halo", "fused", None])
@pytest.mark.parametrize("case", mg.CASES, ids=[c[0] for c in mg.CASES])
def test_shim_bound_reference_reproduces_the_fixtures(case, mode):
    """liblbmi behind lb_collide / lb_halo / lb_propagation / lb_memcpy of the
    reference, in every LBMI_MODE: the driver's dumps after the first collide,
    halo, propagation and after all steps equal the fixtures (the copies to
    the host go through the binding's flush). None: no LBMI_* variable set --
    fused, rho and u on demand (the dumps of hydro->rho, u come through the
    bound hydro_memcpy, which has them formed first)."""
    exe = _exe(case[1], shim=True)
    env = _env(mode)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_case(case, tmp, exe=exe, env=env)
    _compare(out, load_golden(case[0]), case[3])


@pytest.mark.parametrize("case", mg.VISC_CASES, ids=[c[0] for c in mg.VISC_CASES])
def test_shim_with_viscosity_model(case):
    exe = _exe(case[1], shim=True)
    env = _env(None)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_case(case, tmp, visc=1, exe=exe, env=env)
    _compare(out, load_golden(case[0]), case[3])


WALLS = mg.WALL_CASES + mg.SLIP_CASES


@pytest.mark.parametrize("mode", ["eager", "halo", None])
@pytest.mark.parametrize("case", WALLS, ids=[c[0] for c in WALLS])
def test_shim_wall_bbl(case, mode):
    """wall_bbl of the binding -- no-slip with moving walls, partial slip, the
    MAP_COLLOID branch -- on the reference's own device link arrays, between
    the binding's lb_halo and lb_propagation."""
    nvel = case[1]
    exe = _exe(nvel, shim=True)
    env = _env(mode)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_wall_case(case, tmp, exe=exe, env=env)
    g = load_golden(case[0])
    for key in ("status", "linki", "linkj", "linkp", "linku"):
        assert np.array_equal(out[key], g[key])
    q = nvel - g["linkp"]
    a = out["f_bbl"].reshape(nvel, -1)[q, g["linkj"]]
    b = g["f_bbl"].reshape(nvel, -1)[q, g["linkj"]]
    assert np.max(np.abs(a - b)) < 1e-15
    fl = (g["status"] == 0)[1:-1, 1:-1, 1:-1]
    assert relmax(interior(out["f_final"], 1)[:, fl], interior(g["f_final"], 1)[:, fl]) < 1e-12
    import json
    fnet = np.array(json.loads(str(out["meta"]))["fnet"])
    ref = np.array(g["meta"]["fnet"])
    assert np.max(np.abs(fnet - ref)) < 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", mg.IO_CASES, ids=[c[0] for c in mg.IO_CASES])
def test_shim_lb_io_write_and_read(case, tmp_path):
    """lb_io_write of the binding (records packed on the device, written from
    there) = the files of the reference byte for byte; lb_io_read of the
    binding restores the host copy the caller goes on with."""
    import subprocess
    nvel, n, timestep = case[1], case[2], case[3]
    ndist = case[4] if len(case) > 4 else 1
    exe = _exe(nvel, shim=True)
    env = _env(None, LBMI_REPORT="1")
    out = mg.run_io_case(case, str(tmp_path), exe=exe, env=env)
    # the binding wrote them, not the original it can hand back to
    assert _re.search(r"liblbmi report: lb_io_write\s+1\s+0\b", str(out["stderr"])), out["stderr"]
    g = np.load(os.path.join(HERE, "golden", case[0] + ".npz"))
    assert str(out["metadata"]) == str(g["metadata"])
    assert out["data"].tobytes() == g["data"].tobytes()
    d = os.path.join(str(tmp_path), case[0])
    variant = case[5] if len(case) > 5 else None
    ascii_ = (variant == "ascii")                         # distribution_io_format ascii
    if variant == "single":                               # no i/o mode named: old-style files
        assert str(out["meta_text"]) == str(g["meta_text"])
    r = subprocess.run([exe, "ioread", d, *map(str, n), str(timestep)]
                       + ([str(ndist)] if (ndist != 1 or variant) else [])
                       + ([variant] if variant else []), check=True, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert _re.search(r"liblbmi report: lb_io_read\s+1\s+0\b", r.stderr), r.stderr
    back = np.fromfile(os.path.join(d, "readback.f.f64"), dtype="<f8").reshape(g["f0"].shape)
    if ascii_:
        # sixteen significant digits in the text
        assert relmax(interior(back, 1), interior(g["f0"], 1)) < 1e-15
    else:
        assert np.array_equal(interior(back, 1), interior(g["f0"], 1))


def _compare_fe(out, g):
    assert np.array_equal(interior(out["phi"], 1), interior(g["phi"], 1))
    # gradients: valid on the interior (nextra = nhalo - 1 = 0 here)
    assert relmax(interior(out["grad"], 1), interior(g["grad"], 1)) < 1e-12
    assert relmax(interior(out["delsq"], 1), interior(g["delsq"], 1)) < 1e-12
    for key in ("f_collide", "u", "f_final"):
        assert relmax(interior(out[key], 1), interior(g[key], 1)) < 1e-12, key


@pytest.mark.parametrize("mode", ["eager", "halo", None])
@pytest.mark.parametrize("case", mg.BINARY_CASES, ids=[c[0] for c in mg.BINARY_CASES])
def test_shim_two_distribution_step(case, mode):
    """free_energy symmetric_lb inside the reference: phi_lb_to_field,
    field_halo, field_grad_compute, hydro_u_zero, lb_collide (binary), lb_halo,
    lb_propagation -- all through the binding (halo: the propagation of both
    distributions is folded into the next collision; None = no LBMI_* variable
    set: fused, the halo swap of both deferred as well)."""
    exe = _exe(case[1], shim=True)
    env = _env(mode)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_binary_case(case, tmp, exe=exe, env=env)
    _compare_fe(out, load_golden(case[0]))


@pytest.mark.parametrize("mode", ["eager", "halo", None])
@pytest.mark.parametrize("case", mg.RELAX_CASES, ids=[c[0] for c in mg.RELAX_CASES])
def test_shim_stress_relaxation(case, mode):
    """lb_collide with fe->use_stress_relaxation (symmetric free energy)."""
    exe = _exe(case[1], shim=True)
    env = _env(mode)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_relax_case(case, tmp, exe=exe, env=env)
    _compare_fe(out, load_golden(case[0]))


# --- the application itself: the reference's main.c / ludwig.c main loop --------

import json as _json                                         # noqa: E402
import re as _re                                             # noqa: E402
import subprocess as _sp                                     # noqa: E402

INPUTS = os.path.join(HERE, "golden", "inputs")


def _ludwig(inp, mode, shim=True, extra_env=None, nvel=19):
    """Run the reference's executable on an input of tests/golden/inputs.
    mode = None leaves LBMI_MODE unset (the binding's default). Whatever the
    child wrote is kept when it fails (a GPU fault must leave evidence): in
    gpurun_out/ (merged back from the GPU box) and in the assertion message."""
    exe = os.path.join(REF, "ludwig_hip_d3q%d" % nvel + ("_shim" if shim else ""))
    if not os.path.exists(exe):
        # (not a skip: the binaries travel with the snapshot, and a missing one
        # would silently take the whole boundary out of the suite)
        pytest.fail("oracle/_ref/%s is missing: build it in the development "
                    "container with `make -C oracle hip`" % os.path.basename(exe))
    env = dict(os.environ)
    env.pop("LBMI_MODE", None)
    if mode is not None:
        env["LBMI_MODE"] = mode
    env.pop("LBMI_FE", None)
    env.pop("LBMI_HYDRO", None)
    env.update(extra_env or {})
    import shutil
    with tempfile.TemporaryDirectory() as tmp:
        # main.c: the input file is "input" in the working directory
        shutil.copy(os.path.join(INPUTS, inp), os.path.join(tmp, "input"))
        r = _sp.run([exe], cwd=tmp, env=env, capture_output=True, text=True,
                    timeout=600)
    if r.returncode != 0 or "Ludwig finished normally." not in r.stdout:
        tail = "exit code %d\n--- stdout (tail) ---\n%s\n--- stderr (tail) ---\n%s" % (
            r.returncode, r.stdout[-3000:], r.stderr[-3000:])
        keep = os.path.join(HERE, "..", "gpurun_out")
        os.makedirs(keep, exist_ok=True)
        if os.path.isdir(keep):
            with open(os.path.join(keep, "ludwig_failed_%s_%s.log" % (inp, mode)), "w") as fh:
                fh.write("exit code %d\n--- stdout ---\n%s\n--- stderr ---\n%s"
                         % (r.returncode, r.stdout, r.stderr))
        pytest.fail("%s (LBMI_MODE=%s) did not finish: %s" % (inp, mode, tail))
    return r.stdout


def _floats(line):
    return [float(x) for x in _re.findall(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", line)]


def _last(log, tag):
    lines = [l for l in log.splitlines() if l.startswith(tag)]
    assert lines, tag
    return _floats(lines[-1])


@pytest.mark.parametrize("mode", ["eager", "halo", None])
@pytest.mark.parametrize("name", ["dist_1dp", "dist_3du"])
def test_ludwig_application_with_the_binding(name, mode):
    """The reference's executable (main.c, ludwig.c main loop, its input
    parser, initialisation and statistics) built for gfx950 with the binding:
    ten steps of the regression serial-dist-1dp / -3du; the statistics it
    prints are those of the reference's own log, digit for digit."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_d3q19_short.json")))
    ref = ref["serial-dist-" + name[-3:]]["final"]
    log = _ludwig(name + ".inp", mode)
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho_total"] and rho[1] == ref["rho_mean"]
    assert abs(rho[2] - ref["rho_var"]) <= 1e-7 * abs(ref["rho_var"]) + 1e-20
    assert abs(rho[3] - ref["rho_min"]) < 2e-11 and abs(rho[4] - ref["rho_max"]) < 2e-11
    g = _last(log, "[total   ]")
    for a, b in zip(g, ref["momentum"]):
        assert abs(a - b) <= 1e-7 * abs(b) + 1e-12
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        if key not in ref:
            continue
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-8 * abs(b) + 1e-16


@pytest.mark.parametrize("mode", ["eager", "halo", None])
def test_ludwig_binary_fluid_droplet_with_the_binding(mode):
    """free_energy symmetric (finite difference, 27-point gradients, second
    order advection): one step of the relaxing droplet serial-symm-dr1. The
    LB step, hydro_u_zero / hydro_f_zero, field_halo and field_grad_compute
    run through the binding, the force and the Cahn-Hilliard update are the
    reference's own kernels."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_symmetric_drop.json")))
    ref = ref["serial-symm-dr1"]["reports"]["1"]
    log = _ludwig("symm_dr1.inp", mode)
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho_total"]
    assert abs(rho[3] - ref["rho_min"]) < 2e-11 and abs(rho[4] - ref["rho_max"]) < 2e-11
    phi = _last(log, "[phi]")
    assert abs(phi[0] - ref["phi_total"]) < 0.02 and abs(phi[2] - ref["phi_var"]) < 2e-8
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-8 * abs(b) + 1e-16


@pytest.mark.parametrize("mode", ["eager", "halo", "fused", None])
@pytest.mark.parametrize("name", ["spin_lb1", "symm_dr2", "spin_fd1", "symm_pat"])
def test_ludwig_application_more_regressions(name, mode):
    """spin_lb1: free_energy symmetric_lb (two distributions, ghost modes
    off), ten steps of a spinodal quench -- phi_lb_to_field, the binary
    collision, halo and propagation of both distributions through the
    binding. symm_dr2: the droplet with an Arrhenius viscosity model -- the
    collision takes its relaxation times from hydro->eta. spin_fd1: spinodal
    quench with the finite-difference order parameter, ten steps, ghost modes
    off. symm_pat: one step from a patchy phi (steep gradients, large forces).
    Statistics of the reference's logs serial-<name>.log."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))[name]
    log = _ludwig(name + ".inp", mode)
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho"][0]
    # (the variance is a difference of nearly equal sums: absolute, as the
    # reference's own comparison, tests/awk-fp-diff.sh)
    assert abs(rho[2] - ref["rho"][2]) <= 1e-12
    assert abs(rho[3] - ref["rho"][3]) < 2e-11 and abs(rho[4] - ref["rho"][4]) < 2e-11
    phi = _last(log, "[phi]")
    for a, b in zip(phi, ref["phi"]):
        assert abs(a - b) <= 2e-7 * abs(b) + 1e-12
    fed = _last(log, "[fed]")
    assert abs(fed[-1] - ref["fed"]) <= 1e-9 * abs(ref["fed"])
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-16


@pytest.mark.parametrize("mode", ["eager", "halo", None, "fused"])
def test_ludwig_duct_flow_between_walls(mode):
    """serial-rect-ct1: a 1 x 62 x 30 duct with walls in y and z, driven by a
    body force, 100 steps: lb_collide, lb_halo, wall_bbl (on copies of the
    reference's host link arrays, momentum into wall->target->fnet) and
    lb_propagation through the binding, the reference's own
    wall_set_wall_distributions in between; fluid and wall momentum and the
    velocity extrema of the reference's log. fused, and None (LBMI_MODE unset:
    the binding starts in fused by itself): the binding notices the wall links
    at the first wall_set_wall_distributions and continues in halo mode.
    (Round 1 skipped this case after one GPU memory fault; the cause --
    lb->target->param never uploaded, so wall_setu_kernel wrote in front of
    f -- is in CHANGELOG.md and tests/test_duct_fault_replay.py.)"""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))["rect_ct1"]
    log = _ludwig("rect_ct1.inp", mode)
    assert ("execution mode fused -> halo" in log) == (mode in ("fused", None))
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho"][0]
    assert abs(rho[3] - ref["rho"][3]) < 2e-11 and abs(rho[4] - ref["rho"][4]) < 2e-11
    for tag, key in (("[total   ]", "momentum_total"), ("[fluid   ]", "momentum_fluid"),
                     ("[walls   ]", "momentum_walls")):
        a, b = _last(log, tag)[0], ref[key][0]          # x: the driven direction
        assert abs(a - b) <= 2e-7 * abs(b), tag
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-14


@pytest.mark.parametrize("mode", ["eager", "halo", None])
def test_ludwig_droplet_twenty_coupled_steps(mode):
    """d3q19-io/iodrop-mpi1-io1: twenty coupled steps of the relaxing droplet
    (the free-energy force and the Cahn-Hilliard update are the reference's
    kernels; the LB step, the field halos, the gradients and the hydro
    housekeeping go through the binding): the report after step 20."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_symmetric_drop.json")))
    ref = ref["iodrop-mpi1-io1"]["reports"]["20"]
    log = _ludwig("iodrop.inp", mode)
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho_total"]
    assert abs(rho[2] - ref["rho_var"]) <= 1e-12
    assert abs(rho[3] - ref["rho_min"]) < 2e-11 and abs(rho[4] - ref["rho_max"]) < 2e-11
    phi = _last(log, "[phi]")
    assert abs(phi[2] - ref["phi_var"]) <= 2e-7 * ref["phi_var"]
    assert abs(phi[3] - ref["phi_min"]) <= 2e-7 and abs(phi[4] - ref["phi_max"]) <= 2e-7
    assert abs(_last(log, "[fed]")[-1] - ref["fed"]) <= 1e-9 * abs(ref["fed"])
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-16


@pytest.mark.parametrize("mode", ["eager", "halo", "fused"])
def test_ludwig_two_dimensional_lattice(mode):
    """serial-dist-2kh: a 64 x 64 x 1 lattice (one site, three with the halo,
    along the contiguous direction): Kelvin-Helmholtz initial condition, ten
    steps."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))["dist_2kh"]
    log = _ludwig("dist_2kh.inp", mode)
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho"][0]
    assert abs(rho[2] - ref["rho"][2]) <= 1e-12
    assert abs(rho[3] - ref["rho"][3]) < 2e-11 and abs(rho[4] - ref["rho"][4]) < 2e-11
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-16


# --- rows a17 and f2 inside the application ---------------------------------------

def _droplet_checks(log, ref):
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho_total"]
    assert abs(rho[3] - ref["rho_min"]) < 2e-11 and abs(rho[4] - ref["rho_max"]) < 2e-11
    phi = _last(log, "[phi]")
    assert abs(phi[2] - ref["phi_var"]) <= 2e-7 * ref["phi_var"]
    assert abs(phi[3] - ref["phi_min"]) <= 2e-7 and abs(phi[4] - ref["phi_max"]) <= 2e-7
    assert abs(_last(log, "[fed]")[-1] - ref["fed"]) <= 1e-9 * abs(ref["fed"])
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-16


@pytest.mark.parametrize("mode", ["halo", "fused"])
def test_ludwig_droplet_with_the_free_energy_sector_bound(mode):
    """No switch set: phi_force_calculation and phi_cahn_hilliard of the binding
    (one kernel each, no stress array, no flux arrays) in place of the
    reference's pth_* / advection_* / phi_ch_* kernels, on top of everything
    else: twenty coupled steps of the relaxing droplet iodrop-mpi1-io1 (27-point
    gradients, second-order advection), the report after step 20."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_symmetric_drop.json")))
    ref = ref["iodrop-mpi1-io1"]["reports"]["20"]
    log = _ludwig("iodrop.inp", mode)
    assert "phi_force_calculation bound" in log and "phi_cahn_hilliard bound" in log
    _droplet_checks(log, ref)
    # LBMI_FE=0: the reference's own kernels for the free-energy sector
    log = _ludwig("iodrop.inp", mode, extra_env={"LBMI_FE": "0"})
    assert "phi_force_calculation bound" not in log and "phi_cahn_hilliard bound" not in log
    _droplet_checks(log, ref)


@pytest.mark.parametrize("name", ["spin_fd1", "symm_pat"])
def test_ludwig_more_regressions_with_the_free_energy_sector_bound(name):
    """spin_fd1 (spinodal quench, ten steps), symm_pat (ONE step from a patchy
    phi: steep gradients, large forces), no switch set: the handle of the
    binding exists from the lb_memcpy of ludwig.c:507 on, so the first step's
    free-energy sector is bound as well."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))[name]
    log = _ludwig(name + ".inp", None)
    assert "phi_force_calculation bound" in log and "phi_cahn_hilliard bound" in log
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho"][0]
    assert abs(rho[2] - ref["rho"][2]) <= 1e-12
    assert abs(rho[3] - ref["rho"][3]) < 2e-11 and abs(rho[4] - ref["rho"][4]) < 2e-11
    phi = _last(log, "[phi]")
    for a, b in zip(phi, ref["phi"]):
        assert abs(a - b) <= 2e-7 * abs(b) + 1e-12
    fed = _last(log, "[fed]")
    assert abs(fed[-1] - ref["fed"]) <= 1e-9 * abs(ref["fed"])
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-16


def _all(log, tag):
    return [_floats(l) for l in log.splitlines() if l.startswith(tag)]


@pytest.mark.parametrize("inp,steps,folded", [
    # 7-point gradients, first-order advection: the one-kernel form.
    # 32 steps, reports at 10, 20, 30; step 1 runs call by call and arms; the
    # run ENDS on two folded steps (phi, u still in the second arrays when
    # ludwig.c frees its objects)
    ("iodrop7.inp", 32, 32 - 3 - 1),
    # 27-point gradients, second-order advection: lbmi_symmetric_lb_collide
    # runs the single free-energy pass and the collision (no one-kernel form)
    ("iodrop.inp", 20, 20 - 2 - 1),
])
def test_free_energy_sector_folded_into_the_collision(inp, steps, folded):
    """ludwig.c calls field_halo(phi), field_grad_compute,
    phi_force_calculation, phi_cahn_hilliard, hydro_u_zero and lb_collide one
    by one; with nothing set the binding notes the first five of a step that
    follows a qualifying one and runs them inside lb_collide
    (lbmi_symmetric_lb_collide: ONE kernel for BASELINE config 4's scheme) --
    except on steps that report or write, where every array must be what the
    reference has. Every report of the run (rho, phi, the free energy, the
    extrema of u) against the same run with the sector bound call by call
    (LBMI_FE=1), with the reference's own free-energy kernels (LBMI_FE=0) and
    against the reference's executable without the binding."""
    exe = os.path.join(REF, "ludwig_hip_d3q19_shim")
    env = _env(None, LBMI_REPORT="1")
    import shutil
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.join(INPUTS, inp), os.path.join(tmp, "input"))
        r = _sp.run([exe], cwd=tmp, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Ludwig finished normally." in r.stdout, r.stderr[-2000:]
    m = _re.search(r"of (\d+) collisions, rho alone in \d+; free-energy sector folded into (\d+)",
                   r.stderr)
    assert m, r.stderr[-2000:]
    assert int(m.group(1)) == steps and int(m.group(2)) == folded
    assert "free-energy sector folded into lb_collide" in r.stdout
    # every symbol of the sequence was the library's at every step
    for sym in ("field_halo", "field_grad_compute", "phi_force_calculation",
                "phi_cahn_hilliard", "hydro_u_zero", "lb_collide"):
        mm = _re.search(r"liblbmi report: %s\s+(\d+)\s+(\d+)" % sym, r.stderr)
        assert mm and int(mm.group(2)) == 0 and int(mm.group(1)) >= steps, (sym, r.stderr[-1500:])
    log = r.stdout
    others = {"call by call": _ludwig(inp, None, extra_env={"LBMI_FE": "1"}),
              "reference's free-energy kernels": _ludwig(inp, None, extra_env={"LBMI_FE": "0"}),
              "unbound": _ludwig(inp, None, shim=False)}
    assert "folded" not in others["call by call"]
    for who, ref in others.items():
        for tag in ("[rho]", "[phi]", "[fed]", "[minimum ]", "[maximum ]"):
            mine, theirs = _all(log, tag), _all(ref, tag)
            assert len(mine) == len(theirs) and len(mine) >= 2, (who, tag)
            for a, b in zip(mine, theirs):
                assert len(a) == len(b)
                for x, y in zip(a, b):
                    # (eight printed digits; u of 1e-5: the last of them)
                    assert abs(x - y) <= 2e-7 * abs(y) + 1e-13, (who, tag, a, b)


def test_free_energy_sector_folded_d3q27():
    """The same with the D3Q27 build of the reference (k_symm_lb_step<27>):
    every report against the call-by-call binding, the reference's own
    free-energy kernels around the library's collision, and the unbound
    executable."""
    inp = "iodrop7.inp"
    log = _ludwig(inp, None, nvel=27)
    assert "free-energy sector folded into lb_collide" in log
    others = {"call by call": _ludwig(inp, None, extra_env={"LBMI_FE": "1"}, nvel=27),
              "reference's free-energy kernels": _ludwig(inp, None, extra_env={"LBMI_FE": "0"}, nvel=27),
              "unbound": _ludwig(inp, None, shim=False, nvel=27)}
    for who, ref in others.items():
        for tag in ("[rho]", "[phi]", "[fed]", "[minimum ]", "[maximum ]"):
            mine, theirs = _all(log, tag), _all(ref, tag)
            assert len(mine) == len(theirs) and len(mine) >= 2, (who, tag)
            for a, b in zip(mine, theirs):
                for x, y in zip(a, b):
                    assert abs(x - y) <= 2e-7 * abs(y) + 1e-13, (who, tag, a, b)


def test_the_free_energy_binding_leaves_other_cases_to_the_reference():
    """A case outside the conditions of the free-energy binding
    (symmetric_lb: two distributions, no finite-difference order parameter):
    nothing of that sector is bound, the log is the reference's."""
    ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))["spin_lb1"]
    log = _ludwig("spin_lb1.inp", None)
    assert "phi_force_calculation bound" not in log and "phi_cahn_hilliard bound" not in log
    rho = _last(log, "[rho]")
    assert rho[0] == ref["rho"][0] and abs(rho[2] - ref["rho"][2]) <= 1e-12


def _policy(inp, extra_env=None):
    """(mode at exit, who chose it, lazy collisions, collisions) of an
    application run, from the LBMI_REPORT=1 line of the binding."""
    exe = os.path.join(REF, "ludwig_hip_d3q19_shim")
    env = _env(None, LBMI_REPORT="1", **(extra_env or {}))
    import shutil
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.join(INPUTS, inp), os.path.join(tmp, "input"))
        r = _sp.run([exe], cwd=tmp, env=env, capture_output=True, text=True,
                    timeout=600)
    assert r.returncode == 0 and "Ludwig finished normally." in r.stdout, r.stderr[-2000:]
    m = _re.search(r"liblbmi report: execution mode (\w+) \(([^)]*)\); rho, u on "
                   r"demand in (\d+) of (\d+) collisions", r.stderr)
    assert m, r.stderr[-2000:]
    return m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), r.stdout


@pytest.mark.parametrize("inp,mode,lazy", [
    ("dist_3du.inp", "fused", "all"),    # plain single fluid: the headline path
    ("rect_ct1.inp", "halo", "all"),     # wall links: halo; nobody reads rho, u on the device
    ("symm_dr1.inp", "fused", "none"),   # free energy: advection reads u every step
    ("symm_dr2.inp", "fused", "none"),   # + a viscosity model
    ("spin_lb1.inp", "fused", "none"),   # two distributions
])
def test_an_unconfigured_run_is_fused_and_lazy_until_a_consumer_shows(inp, mode, lazy):
    """No LBMI_* variable but LBMI_REPORT: the binding starts every run in
    fused with rho, u on demand and demotes itself per consumer it detects
    (walls -> halo; free energy, viscosity model, two distributions -> rho, u
    stored by every collision). The statistics are checked by the application
    tests above with the same environment; here: what the run ended up as."""
    got_mode, who, nlazy, ncollide, _ = _policy(inp)
    assert who == "chosen by the binding"
    assert got_mode == mode
    assert ncollide > 0
    assert nlazy == (ncollide if lazy == "all" else 0)


def test_lbmi_hydro_store_overrides_the_default():
    got_mode, who, nlazy, ncollide, _ = _policy("dist_3du.inp", {"LBMI_HYDRO": "store"})
    assert got_mode == "fused" and nlazy == 0 and ncollide > 0


@pytest.mark.parametrize("mode", ["halo", "fused"])
@pytest.mark.parametrize("name", ["dist_1dp", "rect_ct1"])
def test_ludwig_with_lazy_hydro(name, mode):
    """LBMI_HYDRO=lazy: lb_collide does not store hydro->rho, u; the velocity
    statistics (hydro_memcpy, then stats_velocity_minmax on the host) get
    them formed from the post-collision distributions: the extrema of the
    reference's logs (Poiseuille start, ten steps; the duct between walls,
    a hundred steps with hydro_u_zero every step and solid sites at rest)."""
    if name == "dist_1dp":
        ref = _json.load(open(os.path.join(HERE, "golden", "regression_d3q19_short.json")))
        ref = ref["serial-dist-1dp"]["final"]
        rho_ref = (ref["rho_total"], ref["rho_min"], ref["rho_max"])
    else:
        ref = _json.load(open(os.path.join(HERE, "golden", "regression_app_extra.json")))[name]
        rho_ref = (ref["rho"][0], ref["rho"][3], ref["rho"][4])
    log = _ludwig(name + ".inp", mode, extra_env={"LBMI_HYDRO": "lazy"})
    rho = _last(log, "[rho]")
    assert rho[0] == rho_ref[0]
    assert abs(rho[3] - rho_ref[1]) < 2e-11 and abs(rho[4] - rho_ref[2]) < 2e-11
    for tag, key in (("[minimum ]", "u_min"), ("[maximum ]", "u_max")):
        for a, b in zip(_last(log, tag), ref[key]):
            assert abs(a - b) <= 2e-7 * abs(b) + 1e-14


# --- the reference's regression suite: runs the binding must hand back ----------

SWEEP = os.path.join(HERE, "golden", "regression_d3q19_short")   # the nine cases tests read


def _sweep_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "regression_sweep", os.path.join(HERE, "..", "tools", "regression_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name,tol", [
    ("serial-chol-st3", 1e-12),   # Lees-Edwards planes + liquid crystal: hydro arrays with buffer planes
    ("serial-le3d-st1", 1e-12),   # Lees-Edwards planes + symmetric free energy
    ("serial-anch-wn1", 1e-12),   # quasi-two-dimensional box: a field halo wider than the box
    ("serial-drop-lc3", 1e-12),   # stress relaxation of a free energy the library does not cover
    ("serial-coll-st1", 2e-11),   # a colloid (bounce_back_on_links on the reference's side);
                                  # [total] momentum: a sum that cancels to round-off
])
def test_ludwig_runs_the_binding_leaves_to_the_reference(name, tol):
    """Inputs of the reference's d3q19-short regression suite that use what
    the library does not cover: the bound executable must print what the
    unbound one prints on the same GPU (tools/regression_sweep.py is the
    whole suite, profiles/r02_regression_sweep.txt its table). Each of the
    first four was a defect of the binding found by that sweep."""
    rs = _sweep_tool()
    env = dict(os.environ)
    for k in ("LBMI_MODE", "LBMI_FE", "LBMI_HYDRO"):
        env.pop(k, None)
    logs = {}
    for leg, exe in (("unbound", "ludwig_hip_d3q19"), ("bound", "ludwig_hip_d3q19_shim")):
        path = os.path.join(REF, exe)
        if not os.path.exists(path):
            pytest.fail("oracle/_ref/%s is missing: `make -C oracle hip`" % exe)
        with tempfile.TemporaryDirectory() as tmp:
            rc, out, err, _ = rs.run_one(name, path, env, tmp, 300)
        assert rc == 0 and "Ludwig finished normally." in out, (leg, rc, out[-2000:], err[-2000:])
        logs[leg] = out
    bad, worst, first = rs.compare(logs["unbound"], logs["bound"], tol)
    assert bad == 0, (worst, first)
    # and both are the run the reference's authors logged (the HIP target moves
    # the last printed digit of some extrema by itself: 8 digits)
    expected = open(os.path.join(SWEEP, name + ".log")).read()
    bad, worst, first = rs.compare(expected, logs["bound"], 5e-8)
    assert bad == 0, (worst, first)


# --- the reference's own unit tests against the binding --------------------------

@pytest.mark.parametrize("nvel", [19, 27])
def test_reference_unit_suites_with_the_binding(nvel):
    """tests/unit/test_model.c, test_halo.c, test_prop.c, test_wall.c of the
    reference (compiled where they lie by oracle/Makefile, assertions on,
    run by oracle/unit_main.c) linked against the binding: every check its
    authors wrote for lb_halo / lb_propagation / lb_memcpy holds for the bound
    functions, and the report shows that the library -- not the original --
    served them (tools/unit_suites.sh: all fifteen suites, both builds)."""
    exe = os.path.join(REF, "unit_hip_d3q%d_shim" % nvel)
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/%s is missing: `make -C oracle hip`" % os.path.basename(exe))
    env = dict(os.environ, LBMI_REPORT="1")
    env.pop("LBMI_MODE", None)
    suites = ["model", "halo", "prop", "wall", "lb_bc_inflow_rhou", "lb_bc_outflow_rhou"]
    with tempfile.TemporaryDirectory() as tmp:
        r = _sp.run([exe] + suites, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    for s in suites:
        assert "DONE     " + s in r.stdout, (s, r.stdout[-2000:], r.stderr[-2000:])
    assert "Failed test assertion" not in r.stdout
    assert r.stdout.count("PASS") >= len(suites)
    calls = {}
    for line in r.stderr.splitlines():
        w = line.split()
        if line.startswith("liblbmi report:") and len(w) == 5 and w[3].isdigit():
            calls[w[2]] = (int(w[3]), int(w[4]))
    # test_prop.c: every lb_halo and lb_propagation is the library's
    assert calls["lb_propagation"][0] >= 4 and calls["lb_propagation"][1] == 0
    # test_halo.c: the device scheme is the library's, the host schemes the original's
    assert calls["lb_halo_swap"][0] >= 16 and calls["lb_halo_swap"][1] >= 1


@pytest.mark.parametrize("name,ncollide", [
    ("serial-auto-c02", 40),      # a Brownian colloid: fluctuations + bounce-back on links
    ("serial-spin-n02", 10),      # spinodal quench, finite-difference order parameter, noise in both
    ("serial-wall-st1", 10),      # fluctuating fluid between walls
])
def test_ludwig_fluctuating_runs_with_the_binding(name, ncollide):
    """isothermal_fluctuations on: the reference's generator states
    (noise->target->state) are advanced by the library's collision exactly as
    the original advances them -- otherwise no statistic of these logs would
    survive ten steps -- and LBMI_REPORT shows that every collision was the
    library's."""
    rs = _sweep_tool()
    env = dict(os.environ, LBMI_REPORT="1")
    for k in ("LBMI_MODE", "LBMI_FE", "LBMI_HYDRO"):
        env.pop(k, None)
    exe = os.path.join(REF, "ludwig_hip_d3q19_shim")
    with tempfile.TemporaryDirectory() as tmp:
        rc, out, err, _ = rs.run_one(name, exe, env, tmp, 300)
    assert rc == 0 and "Ludwig finished normally." in out, (rc, out[-2000:], err[-2000:])
    calls = {}
    for line in err.splitlines():
        w = line.split()
        if line.startswith("liblbmi report:") and len(w) == 5 and w[3].isdigit():
            calls[w[2]] = (int(w[3]), int(w[4]))
    assert calls["lb_collide"] == (ncollide, 0), calls
    expected = open(os.path.join(SWEEP, name + ".log")).read()
    bad, worst, first = rs.compare(expected, out, 5e-8)
    assert bad == 0, (worst, first)
