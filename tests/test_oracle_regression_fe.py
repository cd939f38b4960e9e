"""Pin the oracle's complete binary-fluid step (BASELINE config 4: symmetric
free energy by finite differences + single-distribution LB) against the
reference's own regression logs: a relaxing droplet, 27-point gradients,
second-order advection (CPU)."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.regression_cases import (check_drop_report, drop_phi, drop_report,
                                    load_expected_drop, rest_f)


def oracle_binary_fluid(case, report_at):
    """The time step of ludwig.c:530-860 for this set-up; yields a report
    at the listed steps (the gradients in a report are those of the start
    of the step, as in ludwig_report_statistics ludwig.c:2313-2347)."""
    h = 2
    m = lbo.model(19)
    p = lbo.make_param(19, case["size"], h, "m10", case["eta"], case["zeta"])
    a, b, kappa, mob = case["a"], case["b"], case["kappa"], case["mobility"]
    phi = drop_phi(case, h)
    f = rest_f(m, phi.shape, h)
    fp = np.zeros_like(f)
    u = np.zeros((3,) + phi.shape)
    rho = np.zeros(phi.shape)
    out = {}
    for step in range(1, max(report_at) + 1):
        force = np.zeros((3,) + phi.shape)                # hydro_f_zero
        lbo.field_halo(p, phi, 2)
        grad, delsq = lbo.grad(p, phi, case["grad_npt"])
        lbo.symm_force(p, a, b, kappa, phi, grad, delsq, force)
        lbo.field_halo(p, u, 1)                           # in phi_cahn_hilliard
        lbo.cahn_hilliard(p, a, b, kappa, mob, phi, delsq, u,
                          order=case["advection_order"])
        u[...] = 0.0                                      # hydro_u_zero
        f, fp = lbo.step(p, f, fp, force, None, rho, u)
        if step in report_at:
            out[step] = drop_report(case, phi, grad, lbo.moments(p, f), u, h)
    return out


def test_drop_initial_state():
    for name, case in load_expected_drop().items():
        if name.startswith("_"):
            continue
        phi = drop_phi(case)
        p = lbo.make_param(19, case["size"], 2)
        lbo.field_halo(p, phi, 2)
        grad, _ = lbo.grad(p, phi, case["grad_npt"])
        rep = drop_report(case, phi, grad, None, None)
        check_drop_report(rep, case["initial"])


@pytest.mark.parametrize("name", ["iodrop-mpi1-io1", "serial-symm-dr1"])
def test_drop_regression_log(name):
    case = load_expected_drop()[name]
    steps = sorted(int(k) for k in case["reports"])
    reps = oracle_binary_fluid(case, steps)
    for k in steps:
        check_drop_report(reps[k], case["reports"][str(k)])
