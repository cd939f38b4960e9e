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


# ---- symmetric free energy: the drop regressions (row f2) -------------------


def load_expected_drop():
    with open(os.path.join(HERE, "golden", "regression_symmetric_drop.json")) as fp:
        return json.load(fp)


def drop_phi(case, nhalo=2):
    """field_phi_init_drop (field_phi_init.c:38-82), serial, not centred:
    phi = tanh((r - radius)/xi), r from (L/2, L/2, L/2), xi the interfacial
    width sqrt(-2 kappa / a) (symmetric.c: fe_symm_interfacial_width)."""
    n = case["size"]
    xi = np.sqrt(-2.0 * case["kappa"] / case["a"])
    ax = [np.arange(1, n[a] + 1, dtype=np.float64) - 0.5 * n[a] for a in range(3)]
    x, y, z = np.meshgrid(*ax, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    phi = np.zeros(tuple(m + 2 * nhalo for m in n))
    h = nhalo
    phi[h:-h, h:-h, h:-h] = 1.0 * np.tanh((1.0 / xi) * (r - case["radius"]))
    return phi


def rest_f(model, nall, nhalo):
    """rho = 1, u = 0: f_p = w_p at the interior sites."""
    f = np.zeros((model["nvel"],) + tuple(nall))
    h = nhalo
    for p in range(model["nvel"]):
        f[p, h:-h, h:-h, h:-h] = model["wv"][p]
    return f


def drop_report(case, phi, grad, mom, u, nhalo=2):
    """The quantities ludwig_report_statistics prints for this set-up
    (stats_distribution.c:55-117, cahn_hilliard_stats.c:96-110,
    stats_free_energy.c:115-122 with fe_symm_fed symmetric.c:285-299,
    stats_velocity.c): interior sites only."""
    h = nhalo
    s = (slice(h, -h),) * 3
    ph = phi[s]
    vol = float(ph.size)
    g2 = sum(grad[a][s] ** 2 for a in range(3))
    fed = (0.5 * case["a"] + 0.25 * case["b"] * ph * ph) * ph * ph \
        + 0.5 * case["kappa"] * g2
    out = {"phi_total": ph.sum(), "phi_mean": ph.sum() / vol,
           "phi_var": (ph * ph).sum() / vol - (ph.sum() / vol) ** 2,
           "phi_min": ph.min(), "phi_max": ph.max(), "fed": fed.sum() / vol}
    if mom is not None:
        # stats_distribution_print: rho = sum_p f_p of the propagated
        # distributions (mom = the moments vector: volume, sum rho,
        # sum rho^2, min, max, momentum)
        mean = mom[1] / mom[0]
        out.update({"rho_total": mom[1], "rho_mean": mean,
                    "rho_var": mom[2] / mom[0] - mean * mean,
                    "rho_min": mom[3], "rho_max": mom[4],
                    "u_min": [u[a][s].min() for a in range(3)],
                    "u_max": [u[a][s].max() for a in range(3)]})
    return out


def check_drop_report(rep, exp):
    """Each printed number to its printed precision."""
    digits = {"rho_total": ("dp", 2), "rho_mean": ("dp", 11),
              "rho_min": ("dp", 11), "rho_max": ("dp", 11),
              "rho_var": ("sig", 8), "phi_total": ("sig", 8),
              "phi_mean": ("sig", 8), "phi_var": ("sig", 8),
              "phi_min": ("sig", 8), "phi_max": ("sig", 8),
              "fed": ("sig", 11)}
    for key, ref in exp.items():
        if key in ("u_min", "u_max"):
            for a in range(3):
                assert close_as_printed(rep[key][a], ref[a], sig=8), (key, a, rep[key][a], ref[a])
            continue
        kind, nd = digits[key]
        ok = (close_as_printed(rep[key], ref, nd) if kind == "dp"
              else close_as_printed(rep[key], ref, sig=nd))
        assert ok, (key, rep[key], ref)
