"""Pin the oracle against the reference's own regression logs (CPU)."""

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import interior
from tests.regression_cases import close_as_printed, initial_f, load_expected


@pytest.mark.parametrize("name", ["serial-dist-1dp", "serial-dist-3du"])
def test_regression_log(name):
    case = load_expected()[name]
    m = lbo.model(case["nvel"])
    p = lbo.make_param(case["nvel"], case["size"], 1, case["scheme"],
                       case["eta"], case["zeta"])
    f = initial_f(case, m)
    fp = np.zeros_like(f)
    rho = np.zeros(f.shape[1:])
    u = np.zeros((3,) + f.shape[1:])

    if "initial" in case:
        mo = lbo.moments(p, f)
        assert close_as_printed(mo[1], case["initial"]["rho_total"], 2)
        assert close_as_printed(mo[5], case["initial"]["momentum"][0], sig=8)

    for _ in range(case["steps"]):
        f, fp = lbo.step(p, f, fp, None, None, rho, u)

    exp = case["final"]
    mo = lbo.moments(p, f)
    mean = mo[1] / mo[0]
    var = abs(mo[2] / mo[0] - mean * mean)
    assert close_as_printed(mo[1], exp["rho_total"], 2)
    assert close_as_printed(mean, exp["rho_mean"], 11)
    assert close_as_printed(mo[3], exp["rho_min"], 11)
    assert close_as_printed(mo[4], exp["rho_max"], 11)
    if exp["rho_var"] > 1e-12:
        assert close_as_printed(var, exp["rho_var"], sig=8)
    else:
        assert var < 1e-12
    for a in range(3):
        ref = exp["momentum"][a]
        if abs(ref) > 1e-6:
            assert close_as_printed(mo[5 + a], ref, sig=8)
        else:
            assert abs(mo[5 + a]) < 1e-9
    if "u_min" in exp:
        ui = interior(u, 1)
        for a in range(3):
            for key, fn in (("u_min", np.min), ("u_max", np.max)):
                ref = exp[key][a]
                if abs(ref) > 1e-12:
                    assert close_as_printed(fn(ui[a]), ref, sig=8)
                else:
                    assert abs(fn(ui[a])) < 1e-12
