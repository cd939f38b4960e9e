#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref).

TEST INFRASTRUCTURE ONLY. Run in the development container, where
/root/reference exists:

    make -C oracle ref && python oracle/make_golden.py

Each fixture holds inputs and the reference's outputs (data only) for one
parameter set of the hot path lb_collide -> lb_halo -> lb_propagation
(reference ludwig.c:802-860), produced by oracle/ref_driver.c linked against
the reference objects built (with assertions on, -DADDR_SOA) from the sources
under /root/reference. Arrays are raw float64 in the reference's SoA order,
shape (nvel, nall_x, nall_y, nall_z) (hydro fields: (3, ...) or (...)).

Keys: meta (JSON string), f0, force, f_collide, rho, u, f_prop, f_final and,
for some cases, f_halo and records (the binary per-site record stream of
lb_io_aggr_pack / lb_write_buf for f_final, shape (nx, ny, nz, nvel)).
"""

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")

# name, nvel, (nx, ny, nz), nhalo, scheme, eta, zeta, fbody, fieldforce,
# solid, nsteps, keep_halo
CASES = [
    ("q19_m10",         19, (6, 5, 4), 1, "m10", 0.1, 0.3, (0, 0, 0), 0, 0, 5, True),
    ("q19_bgk",         19, (6, 5, 4), 1, "bgk", 0.1, 0.1, (0, 0, 0), 0, 0, 5, False),
    ("q19_trt",         19, (6, 5, 4), 1, "trt", 0.05, 0.2, (0, 0, 0), 0, 0, 5, False),
    ("q19_m10_fbody",   19, (6, 5, 4), 1, "m10", 0.1, 0.3, (1e-5, 2e-5, -3e-5), 0, 0, 5, False),
    ("q19_bgk_ffield",  19, (6, 5, 4), 1, "bgk", 0.2, 0.2, (2e-6, 0, 1e-6), 1, 0, 5, False),
    ("q19_trt_ffield",  19, (6, 5, 4), 1, "trt", 0.1, 0.3, (0, 1e-6, 0), 1, 0, 5, False),
    ("q19_m10_nh2_ffield", 19, (6, 5, 4), 2, "m10", 0.1, 0.3, (0, 0, 0), 1, 0, 5, True),
    ("q19_m10_solid",   19, (6, 5, 4), 1, "m10", 0.1, 0.3, (1e-5, 0, 0), 0, 1, 1, False),
    ("q27_m10",         27, (6, 5, 4), 1, "m10", 0.1, 0.3, (0, 0, 0), 0, 0, 5, True),
    ("q27_bgk",         27, (6, 5, 4), 1, "bgk", 0.1, 0.1, (0, 0, 0), 0, 0, 5, False),
    ("q27_m10_ffield",  27, (6, 5, 4), 1, "m10", 0.1, 0.3, (1e-5, -1e-5, 2e-5), 1, 0, 5, False),
    ("q27_bgk_nh2",     27, (5, 4, 6), 2, "bgk", 0.15, 0.15, (0, 0, 1e-5), 0, 0, 5, False),
]

# the same tuple; lb_collide is given a viscosity model, i.e. the local shear
# viscosity comes from hydro->eta (key "eta" in the fixture)
VISC_CASES = [
    ("visc_q19_m10", 19, (6, 5, 4), 1, "m10", 0.1, 0.3, (1e-5, 0, 0), 0, 0, 4, False),
    ("visc_q19_trt", 19, (6, 5, 4), 1, "trt", 0.05, 0.2, (0, 0, 0), 1, 0, 4, False),
    ("visc_q27_bgk", 27, (5, 4, 6), 1, "bgk", 0.15, 0.15, (0, 0, 1e-5), 0, 0, 4, False),
]


# the same tuple + (visc, kt, ghosts): isothermal fluctuations
# (noise->on[NOISE_RHO], temperature kt); keys noise0, noise_final = the
# generator states (4, nall) uint32 before the first and after the last step
NOISE_CASES = [
    (("noise_q19_m10", 19, (6, 5, 4), 1, "m10", 0.1, 0.3, (0, 0, 0), 0, 0, 4, False), 0, 1e-4, 1),
    (("noise_q19_trt_solid", 19, (6, 5, 4), 1, "trt", 0.05, 0.2, (1e-5, 0, 0), 1, 1, 4, False), 0, 2e-4, 1),
    (("noise_q19_bgk_visc", 19, (6, 5, 4), 1, "bgk", 0.2, 0.2, (0, 0, 0), 0, 0, 4, False), 1, 1e-4, 1),
    (("noise_q19_m10_noghost", 19, (5, 4, 6), 1, "m10", 0.1, 0.3, (0, 0, 0), 0, 0, 4, False), 0, 1e-4, 0),
]


def run_case(case, tmp, visc=0, exe=None, env=None, noise=None):
    """exe, env: another build of the same driver (tests/test_gpu_shim.py runs
    the reference's HIP target, with and without the binding, this way)."""
    (name, nvel, n, nhalo, scheme, eta, zeta, fb, ff, solid, nsteps,
     keep_halo) = case
    if exe is None:
        exe = os.path.join(HERE, "_ref", "ref_driver_d3q%d" % nvel)
    prefix = os.path.join(tmp, name)
    args = [exe, "dump", prefix, *map(str, n), str(nhalo), scheme,
            repr(eta), repr(zeta), *[repr(float(x)) for x in fb],
            str(ff), str(solid), str(nsteps)]
    if visc or noise:
        args.append("1" if visc else "0")
    if noise:
        args += [repr(float(noise[0])), str(int(noise[1]))]
    subprocess.run(args, check=True, env=env)
    meta = json.load(open(prefix + ".json"))
    meta["name"] = name
    meta["scheme_name"] = scheme
    nall = tuple(meta["nall"])

    def load(key, lead):
        a = np.fromfile("%s.%s.f64" % (prefix, key), dtype="<f8")
        return a.reshape(lead + nall)

    out = {"meta": np.array(json.dumps(meta))}
    out["f0"] = load("f0", (nvel,))
    out["force"] = load("force", (3,))
    out["f_collide"] = load("f_collide", (nvel,))
    out["rho"] = load("rho", ())
    out["u"] = load("u", (3,))
    if keep_halo:
        out["f_halo"] = load("f_halo", (nvel,))
        rec = np.fromfile("%s.records.f64" % prefix, dtype="<f8")
        out["records"] = rec.reshape(tuple(meta["nlocal"]) + (nvel,))
    out["f_prop"] = load("f_prop", (nvel,))
    out["f_final"] = load("f_final", (nvel,))
    if visc:
        out["eta"] = load("eta", ())
    if noise:
        for key in ("noise0", "noise_final"):
            a = np.fromfile("%s.%s.i32" % (prefix, key), dtype="<u4")
            out[key] = a.reshape((4,) + nall)
    return out


# name, (nx, ny, nz), a, b, kappa, mobility, gradient stencil, advection order
FE_CASES = [
    ("fe_symm_a", (6, 5, 4), -0.0625, 0.0625, 0.04, 0.15, 7, 1),
    ("fe_symm_b", (5, 8, 7), -0.00625, 0.00625, 0.004, 1.25, 7, 1),
    ("fe_symm_c", (6, 5, 4), -0.0625, 0.0625, 0.04, 0.15, 27, 2),
    ("fe_symm_d", (5, 8, 7), -0.00625, 0.00625, 0.004, 1.25, 27, 3),
    ("fe_symm_e", (7, 6, 5), -0.0625, 0.0625, 0.04, 0.15, 7, 4),
]


def run_fe_case(case, tmp):
    """Symmetric free-energy force chain (row f2): phi with its width-2 halo,
    grad/delsq of grad_3d_7pt_fluid or grad_3d_27pt_fluid, force of
    pth_stress_compute + pth_force_fluid_driver; u (with its halo) and
    phi_new of phi_cahn_hilliard (advection of order 1..4)."""
    name, n, a, b, kappa, mobility, gradnpt, advorder = case
    exe = os.path.join(HERE, "_ref", "ref_driver_d3q19")
    prefix = os.path.join(tmp, name)
    subprocess.run([exe, "fe", prefix, *map(str, n), repr(a), repr(b),
                    repr(kappa), repr(mobility), str(gradnpt), str(advorder)],
                   check=True)
    meta = json.load(open(prefix + ".json"))
    meta["name"] = name
    nall = tuple(meta["nall"])

    def load(key, lead):
        return np.fromfile("%s.%s.f64" % (prefix, key), dtype="<f8").reshape(lead + nall)

    return {"meta": np.array(json.dumps(meta)), "phi": load("phi", ()),
            "grad": load("grad", (3,)), "delsq": load("delsq", ()),
            "force": load("force", (3,)), "u": load("u", (3,)),
            "phi_new": load("phi_new", ())}


# name, nvel, (nx, ny, nz), a, b, kappa, mobility, eta, zeta, fbody_x, nsteps
BINARY_CASES = [
    ("bin_q19_a", 19, (6, 5, 4), -0.0625, 0.0625, 0.04, 0.15, 0.1, 0.1, 1e-5, 4),
    ("bin_q19_b", 19, (5, 8, 7), -0.00625, 0.00625, 0.004, 3.75, 0.00625, 0.00625, 0.0, 4),
    ("bin_q27_a", 27, (6, 5, 4), -0.0625, 0.0625, 0.04, 0.5, 0.1, 0.3, 0.0, 4),
]


# fe->use_stress_relaxation with ONE distribution (same arguments)
RELAX_CASES = [
    ("relax_q19_a", 19, (6, 5, 4), -0.0625, 0.0625, 0.04, 0.15, 0.1, 0.1, 1e-5, 4),
    ("relax_q27_a", 27, (5, 6, 4), -0.00625, 0.00625, 0.004, 1.25, 0.1, 0.3, 0.0, 4),
]


def run_relax_case(case, tmp, exe=None, env=None):
    """lb_collide with fe->use_stress_relaxation = 1 (collision.c:413-429):
    the symmetric stress of a fixed phi in the equilibrium stress."""
    name, nvel, n, a, b, kappa, mob, eta, zeta, fx, nsteps = case
    if exe is None:
        exe = os.path.join(HERE, "_ref", "ref_driver_d3q%d" % nvel)
    prefix = os.path.join(tmp, name)
    subprocess.run([exe, "relax", prefix, *map(str, n), repr(a), repr(b),
                    repr(kappa), repr(mob), repr(eta), repr(zeta), repr(fx),
                    str(nsteps)], check=True, env=env)
    meta = json.load(open(prefix + ".json"))
    meta["name"] = name
    nall = tuple(meta["nall"])

    def load(key, lead):
        return np.fromfile("%s.%s.f64" % (prefix, key), dtype="<f8").reshape(lead + nall)

    return {"meta": np.array(json.dumps(meta)), "f0": load("f0", (nvel,)),
            "phi": load("phi", ()), "grad": load("grad", (3,)),
            "delsq": load("delsq", ()), "f_collide": load("f_collide", (nvel,)),
            "rho": load("rho", ()), "u": load("u", (3,)),
            "f_final": load("f_final", (nvel,))}


# the same tuple + kt: the two-distribution step with isothermal fluctuations
NOISE_BINARY_CASES = [
    (("noise_bin_q19_a", 19, (6, 5, 4), -0.0625, 0.0625, 0.04, 0.15, 0.1, 0.1, 1e-5, 4), 1e-4),
]


def run_binary_case(case, tmp, exe=None, env=None, kt=None):
    """The two-distribution (symmetric_lb) step: lb_collision_binary with
    27-point gradients, lb_halo and lb_propagation of both distributions.
    Arrays (2*nvel, nall) are n-major: [0:nvel] density, [nvel:] order
    parameter."""
    name, nvel, n, a, b, kappa, mob, eta, zeta, fx, nsteps = case
    if exe is None:
        exe = os.path.join(HERE, "_ref", "ref_driver_d3q%d" % nvel)
    prefix = os.path.join(tmp, name)
    subprocess.run([exe, "binary", prefix, *map(str, n), repr(a), repr(b),
                    repr(kappa), repr(mob), repr(eta), repr(zeta), repr(fx),
                    str(nsteps)] + ([repr(float(kt))] if kt else []), check=True, env=env)
    meta = json.load(open(prefix + ".json"))
    meta["name"] = name
    nall = tuple(meta["nall"])

    def load(key, lead):
        return np.fromfile("%s.%s.f64" % (prefix, key), dtype="<f8").reshape(lead + nall)

    out = {"meta": np.array(json.dumps(meta)), "f0": load("f0", (2 * nvel,)),
           "phi": load("phi", ()), "grad": load("grad", (3,)),
           "delsq": load("delsq", ()), "f_collide": load("f_collide", (2 * nvel,)),
           "u": load("u", (3,)), "f_final": load("f_final", (2 * nvel,))}
    if kt:
        for key in ("noise0", "noise_final"):
            a = np.fromfile("%s.%s.i32" % (prefix, key), dtype="<u4")
            out[key] = a.reshape((4,) + nall)
    return out


# name, nvel, (nx, ny, nz), isboundary, ubot_y, utop_y, solid block, nsteps
WALL_CASES = [
    ("wall_q19_x", 19, (6, 5, 4), (1, 0, 0), -0.01, 0.02, 0, 4),
    ("wall_q19_xyz", 19, (5, 6, 4), (1, 1, 1), 0.0, 0.0, 1, 4),
    ("wall_q27_z", 27, (5, 4, 6), (0, 0, 1), 0.01, -0.03, 0, 4),
    # solid = 2: MAP_COLLOID marks on fluid sites after the links were built
    ("wall_q19_colloid", 19, (6, 5, 4), (1, 0, 0), -0.01, 0.02, 2, 3),
]


# name, nvel, (nx, ny, nz), isboundary, sbot, stop, nsteps: walls at rest with
# partial slip (wall_bbl_slip_kernel); faces, edges and corners
SLIP_CASES = [
    ("slip_q19_z", 19, (5, 4, 6), (0, 0, 1), (0, 0, 0.5), (0, 0, 1.0), 4),
    ("slip_q19_xyz", 19, (5, 6, 4), (1, 1, 1), (0.5, 0.25, 1.0), (0.0, 0.75, 0.3), 4),
    ("slip_q27_xy", 27, (4, 5, 6), (1, 1, 0), (1.0, 0.4, 0), (0.6, 0.0, 0), 4),
    ("slip_q19_colloid", 19, (5, 4, 6), (0, 0, 1), (0, 0, 0.5), (0, 0, 1.0), 3, 2),
]


def run_wall_case(case, tmp, exe=None, env=None):
    """Flat walls with bounce-back on links: status map, links and f after
    the first wall_bbl and after nsteps of collide, halo, wall_bbl,
    propagation; meta["fnet"] = the accumulated wall momentum."""
    slip = []
    if isinstance(case[4], tuple):
        name, nvel, n, bnd, sbot, stop, nsteps = case[:7]
        ubot = utop = 0.0
        solid = case[7] if len(case) > 7 else 0
        slip = [repr(float(v)) for v in (*sbot, *stop)]
    else:
        name, nvel, n, bnd, ubot, utop, solid, nsteps = case
    if exe is None:
        exe = os.path.join(HERE, "_ref", "ref_driver_d3q%d" % nvel)
    prefix = os.path.join(tmp, name)
    subprocess.run([exe, "wall", prefix, *map(str, n), *map(str, bnd),
                    repr(ubot), repr(utop), str(solid), str(nsteps), *slip],
                   check=True, env=env)
    meta = json.load(open(prefix + ".json"))
    meta["name"] = name
    meta["scheme_name"] = "m10"
    nall = tuple(meta["nall"])

    def load(key, lead):
        return np.fromfile("%s.%s.f64" % (prefix, key), dtype="<f8").reshape(lead + nall)

    def loadi(key):
        return np.fromfile("%s.%s.i32" % (prefix, key), dtype="<i4")

    out = {"meta": np.array(json.dumps(meta)),
           "status": loadi("status").reshape(nall).astype(np.int8),
           "linki": loadi("linki"), "linkj": loadi("linkj"),
           "linkp": loadi("linkp"), "linku": loadi("linku"),
           "f0": load("f0", (nvel,)), "f_bbl": load("f_bbl", (nvel,)),
           "f_final": load("f_final", (nvel,))}
    if slip:
        out.update(linkk=loadi("linkk"), linkq=loadi("linkq"), links=loadi("links"))
    return out


# name, nvel, (nx, ny, nz), timestep
IO_CASES = [
    ("io_q19", 19, (6, 5, 4), 7),
    ("io_q27", 27, (5, 4, 6), 123456),
    ("io_q19_2dist", 19, (4, 6, 5), 40, 2),         # ndist = 2: records of 38 doubles
    ("io_q19_ascii", 19, (6, 4, 3), 12, 1, "ascii"),         # distribution_io_format ascii
    ("io_q19_2dist_ascii", 19, (3, 4, 5), 3, 2, "ascii"),    # two values per line
    # no i/o mode named in the input (io_options_default(): single): the
    # old-style files of io_harness.c
    ("io_q19_single", 19, (5, 6, 4), 20, 1, "single"),
    ("io_q27_2dist_single", 27, (4, 3, 5), 1234567, 2, "single"),
    # not periodic in z (a run with walls there): the metadata says so
    ("io_q19_wallz", 19, (4, 4, 3), 8, 1, "wallz"),
]


def run_io_case(case, tmp, exe=None, env=None):
    """lb_io_write of the reference (MPI-IO mode, one file): the metadata
    file (text), the data file (bytes) and the f they were written from."""
    name, nvel, n, timestep = case[:4]
    ndist = case[4] if len(case) > 4 else 1
    if exe is None:
        exe = os.path.join(HERE, "_ref", "ref_driver_d3q%d" % nvel)
    d = os.path.join(tmp, name)
    os.makedirs(d)
    variant = case[5] if len(case) > 5 else None      # "ascii" | "single" | "wallz"
    r = subprocess.run([exe, "io", d, *map(str, n), str(timestep)]
                       + ([str(ndist)] if (ndist != 1 or variant) else [])
                       + ([variant] if variant else []), check=True,
                       stdout=subprocess.DEVNULL, env=env,
                       stderr=(subprocess.PIPE if env else None), text=True)
    datafile = ("dist-%8.8d.001-001" if variant == "single" else "dist-%9.9d.001-001") % timestep
    nall = tuple(m + 2 for m in n)
    extra = {}
    if env:
        extra["stderr"] = np.array(r.stderr)      # (a test's run, not a fixture)
    if variant == "single":
        extra["meta_text"] = np.array(open(os.path.join(d, "dist.001-001.meta")).read())
    return {**extra,
            "metadata": np.array(open(os.path.join(d, "dist-metadata.001-001")).read()),
            "datafile": np.array(datafile),
            "data": np.fromfile(os.path.join(d, datafile), dtype=np.uint8),
            "f0": np.fromfile(os.path.join(d, "written.f0.f64"),
                              dtype="<f8").reshape((ndist * nvel,) + nall),
            "timestep": np.array(timestep)}


def main():
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for case in CASES:
            out = run_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in VISC_CASES:
            out = run_case(case, tmp, visc=1)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case, visc, kt, ghosts in NOISE_CASES:
            out = run_case(case, tmp, visc=visc, noise=(kt, ghosts))
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in FE_CASES:
            out = run_fe_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in BINARY_CASES:
            out = run_binary_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case, kt in NOISE_BINARY_CASES:
            out = run_binary_case(case, tmp, kt=kt)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in RELAX_CASES:
            out = run_relax_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in WALL_CASES + SLIP_CASES:
            out = run_wall_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
        for case in IO_CASES:
            out = run_io_case(case, tmp)
            fn = os.path.join(GOLD, case[0] + ".npz")
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))
    return 0


if __name__ == "__main__":
    sys.exit(main())
