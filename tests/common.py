"""Shared helpers for the tests: golden fixtures and comparison metrics."""

import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")

# Parity tolerances (BASELINE.json north_star: 1e-12 relative on conserved
# density/momentum; SURVEY.md 8(d): max|df|/max|f| <= 1e-12).
RTOL_F = 1.0e-12
RTOL_CONSERVED = 1.0e-12


def golden_names():
    """Fixtures of the LB step (q19_*, q27_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "q*.npz")))


def golden_fe_names():
    """Fixtures of the symmetric free-energy force chain (fe_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "fe_*.npz")))


def golden_binary_names():
    """Fixtures of the two-distribution (symmetric_lb) step (bin_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "bin_*.npz")))


def golden_visc_names():
    """Fixtures of lb_collide with a viscosity model (visc_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "visc_*.npz")))


def golden_relax_names():
    """Fixtures of lb_collide with fe->use_stress_relaxation (relax_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "relax_*.npz")))


def golden_wall_names():
    """Fixtures of flat walls with bounce-back on links (wall_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "wall_*.npz")))


def golden_slip_names():
    """Fixtures of flat walls with partial slip (slip_*)."""
    return sorted(os.path.basename(f)[:-4]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "slip_*.npz")))


def load_io_golden(name):
    """Files written by the reference's lb_io_write (io_q19, io_q27): the
    metadata text, the data file name and bytes, and the f they hold."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {"metadata": str(z["metadata"]), "datafile": str(z["datafile"]),
         "data": z["data"].tobytes(), "f0": z["f0"],
         "timestep": int(z["timestep"])}
    if "meta_text" in z.files:          # the single mode's dist.001-001.meta
        g["meta_text"] = str(z["meta_text"])
    return g


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta"]))
    return g


def interior(a, nhalo):
    """Interior view of an array whose last three axes are (x, y, z)."""
    h = nhalo
    return a[..., h:-h, h:-h, h:-h]


def xplanes(a, nhalo):
    """Sites of the x-interior planes, with the width-1 y/z halo ring."""
    h = nhalo
    o = nhalo - 1
    return a[..., h:-h, o:a.shape[-2] - o, o:a.shape[-1] - o]


def shell1(a, nhalo):
    """Interior plus the width-1 halo shell next to it.

    This is the region lb_halo() defines (halo_swap.c, nswap = 1). Layers
    further out (nhalo > 1) are not exchanged; in the reference they hold
    whatever its unmasked halo-site collision left there (NaN from 0/0).
    """
    o = nhalo - 1
    if o == 0:
        return a
    return a[..., o:-o, o:-o, o:-o]


def relmax(a, b):
    """max|a-b| / max|b|"""
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def status_from_meta(meta):
    """The MAP_BOUNDARY block of oracle/ref_driver.c:init_map."""
    nall = tuple(meta["nall"])
    st = np.zeros(nall, dtype=np.int8)
    if meta["solid"]:
        h = meta["nhalo"]
        st[h + 1:h + 3, h + 1:h + 3, h + 1:h + 3] = 1
    return st


def momentum_scale(f, cv, nhalo):
    """sum over interior sites and populations of |f_p c_pa| (max over a):
    the scale rounding errors in the total momentum are proportional to.

    The synthetic states have a net momentum that cancels to ~0 (sines over
    whole periods) and even the per-site momentum is ~1e-2 of the population
    sum it is formed from, so |sum g| is an ill-conditioned denominator for a
    relative error; sum |f c| is the condition-free one (it equals |sum g| up
    to a factor ~1/u for a uniform flow).
    """
    fi = np.abs(interior(f, nhalo))
    c = np.abs(cv.astype(np.float64))
    g = np.tensordot(c.T, fi, axes=(1, 0))
    return float(np.max(np.sum(g, axis=(1, 2, 3))))
