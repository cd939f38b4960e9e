"""Inputs of the reference regression cases that pin the hot path.

The initial conditions restate src/distribution_rt.c (lb_init_poiseuille
:540-585, lb_init_uniform :504-533) on top of the second-order equilibrium
(model.c:915-941); expected numbers are in
tests/golden/regression_d3q19_short.json (the reference's own logs).
"""

import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_expected():
    with open(os.path.join(HERE, "golden", "regression_d3q19_short.json")) as fp:
        return json.load(fp)


def equilibrium(model, rho, u):
    """f_p = rho w_p (1 + 3 u.c + 4.5 (cc - 1/3):uu), u shape (3, ...)."""
    cv = model["cv"].astype(np.float64)
    wv = model["wv"]
    shape = (model["nvel"],) + u.shape[1:]
    f = np.zeros(shape)
    for p in range(model["nvel"]):
        udotc = sum(u[a] * cv[p, a] for a in range(3))
        sdotq = 0.0
        for a in range(3):
            for b in range(3):
                sdotq = sdotq + (cv[p, a] * cv[p, b]
                                 - (1.0 / 3.0) * (a == b)) * u[a] * u[b]
        f[p] = rho * wv[p] * (1.0 + 3.0 * udotc + 4.5 * sdotq)
    return f


def initial_f(case, model, nhalo=1):
    n = case["size"]
    nall = tuple(x + 2 * nhalo for x in n)
    u = np.zeros((3,) + tuple(n))
    if case["init"] == "1d_poiseuille":
        # x = (noffset + ic) - lmin, lmin = 0.5; u = umax x (L - x) 4 / L^2
        for a in range(3):
            L = float(n[a])
            x = np.arange(1, n[a] + 1) - 0.5
            prof = case["umax"][a] * x * (L - x) * 4.0 / (L * L)
            shape = [1, 1, 1]
            shape[a] = n[a]
            u[a] = prof.reshape(shape)
    elif case["init"] == "3d_uniform_u":
        for a in range(3):
            u[a] = case["u0"][a]
    else:
        raise ValueError(case["init"])
    fi = equilibrium(model, 1.0, u)
    f = np.zeros((model["nvel"],) + nall)
    h = nhalo
    f[:, h:-h, h:-h, h:-h] = fi
    return f


def close_as_printed(x, ref, digits_after_point=None, sig=None):
    """|x - ref| within half a unit of the last printed digit + 1e-12
    (tests/awk-fp-diff.sh:37 allows 1e-12 absolute on printed tokens)."""
    if sig is not None:
        if ref == 0.0:
            tol = 1e-12
        else:
            tol = 0.5 * 10.0 ** (np.floor(np.log10(abs(ref))) - sig + 1)
    else:
        tol = 0.5 * 10.0 ** (-digits_after_point)
    return abs(x - ref) <= tol + 1e-12
