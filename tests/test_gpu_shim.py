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
        pytest.skip("oracle/_ref/%s not built (make -C oracle hip)" % os.path.basename(exe))
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


@pytest.mark.parametrize("mode", ["eager", "halo", "fused"])
@pytest.mark.parametrize("case", mg.CASES, ids=[c[0] for c in mg.CASES])
def test_shim_bound_reference_reproduces_the_fixtures(case, mode):
    """liblbmi behind lb_collide / lb_halo / lb_propagation / lb_memcpy of the
    reference, in every LBMI_MODE: the driver's dumps after the first collide,
    halo, propagation and after all steps equal the fixtures (the copies to
    the host go through the binding's flush)."""
    exe = _exe(case[1], shim=True)
    env = dict(os.environ, LBMI_MODE=mode)
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_case(case, tmp, exe=exe, env=env)
    _compare(out, load_golden(case[0]), case[3])


@pytest.mark.parametrize("case", mg.VISC_CASES, ids=[c[0] for c in mg.VISC_CASES])
def test_shim_with_viscosity_model(case):
    exe = _exe(case[1], shim=True)
    env = dict(os.environ, LBMI_MODE="fused")
    with tempfile.TemporaryDirectory() as tmp:
        out = mg.run_case(case, tmp, visc=1, exe=exe, env=env)
    _compare(out, load_golden(case[0]), case[3])
