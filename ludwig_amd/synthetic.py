"""The seeded synthetic input of SURVEY.md section 8(d) (host side, numpy).

Second-order equilibrium (reference model.c:915-941) of
    rho = 1 + 0.01 cos(2 pi (x/Lx + y/Ly + z/Lz)),
    u   = 0.01 (sin 2 pi y/Ly, sin 2 pi z/Lz, sin 2 pi x/Lx),
each population multiplied by (1 + 1e-3 (r - 1/2)), r from the 32-bit LCG
s <- 1664525 s + 1013904223 (seed 12345) advanced in global (x, y, z, p)
order. This is the generator that produced tests/golden/*.npz (f0), so small
boxes reproduce the fixtures; bench.py uses it for the 256^3 workload.
"""

import numpy as np

_A = 1664525
_C = 1013904223
_MASK = 0xFFFFFFFF


def _jump(n):
    """(a, c) such that n LCG steps are s -> a s + c (mod 2^32)."""
    a, c = 1, 0
    ab, cb = _A, _C
    while n:
        if n & 1:
            a = (ab * a) & _MASK
            c = (ab * c + cb) & _MASK
        cb = ((ab + 1) * cb) & _MASK
        ab = (ab * ab) & _MASK
        n >>= 1
    return a, c


def lcg_uniform(k0, count):
    """r_k for k = k0 .. k0+count-1, r_k = s_{k+1}/2^32, s_0 = 12345."""
    a, c = _jump(k0 + 1)
    out = np.empty(count, dtype=np.uint32)
    out[0] = (a * 12345 + c) & _MASK
    m = 1
    while m < count:
        am, cm = _jump(m)
        n = min(m, count - m)
        # uint32 arithmetic wraps modulo 2^32
        out[m:m + n] = out[:n] * np.uint32(am) + np.uint32(cm)
        m += n
    return out.astype(np.float64) / 4294967296.0


def equilibrium(cv, wv, rho, u):
    """f_p = rho w_p (1 + 3 u.c + 4.5 (c c - 1/3 delta):uu); u: (3, ...)."""
    nvel = len(wv)
    f = np.empty((nvel,) + rho.shape)
    for p in range(nvel):
        c = cv[p].astype(np.float64)
        udotc = u[0] * c[0] + u[1] * c[1] + u[2] * c[2]
        sdotq = 0.0
        for a in range(3):
            for b in range(3):
                sdotq = sdotq + (c[a] * c[b] - (1.0 / 3.0) * (a == b)) * u[a] * u[b]
        f[p] = rho * wv[p] * (1.0 + 3.0 * udotc + 4.5 * sdotq)
    return f


def x_plane(cv, wv, ntotal, ix):
    """Interior values f[(nvel, ny, nz)] of the global x-plane ix (0-based)."""
    nvel = len(wv)
    nx, ny, nz = ntotal
    y = (np.arange(ny) / ny)[:, None]
    z = (np.arange(nz) / nz)[None, :]
    x = ix / nx
    rho = 1.0 + 0.01 * np.cos(2.0 * np.pi * (x + y + z))
    u = np.empty((3, ny, nz))
    u[0] = 0.01 * np.sin(2.0 * np.pi * y)
    u[1] = 0.01 * np.sin(2.0 * np.pi * z)
    u[2] = 0.01 * np.sin(2.0 * np.pi * x)
    f = equilibrium(cv, wv, rho, u)
    r = lcg_uniform(ix * ny * nz * nvel, ny * nz * nvel).reshape(ny, nz, nvel)
    f *= 1.0 + 1.0e-3 * (np.moveaxis(r, 2, 0) - 0.5)
    return f


def fill(cv, wv, ntotal, nhalo=1, xrange=None, out=None):
    """Host array (nvel, nall_x, nall_y, nall_z) for the slab xrange
    (global 0-based [x0, x1)), halo sites zero."""
    nvel = len(wv)
    x0, x1 = (0, ntotal[0]) if xrange is None else xrange
    h = nhalo
    shape = (nvel, x1 - x0 + 2 * h, ntotal[1] + 2 * h, ntotal[2] + 2 * h)
    if out is None:
        out = np.zeros(shape)
    assert out.shape == shape
    for ix in range(x0, x1):
        out[:, h + ix - x0, h:-h, h:-h] = x_plane(cv, wv, ntotal, ix)
    return out


def fill_device(lb, cv, wv, ntotal, xrange=None):
    """Upload the synthetic state plane by plane into lb.f (torch plumbing)."""
    import torch
    x0, x1 = (0, ntotal[0]) if xrange is None else xrange
    h = lb.nhalo
    lb.lb_flush()
    lb.synchronize()
    f = lb.f
    f.zero_()
    for ix in range(x0, x1):
        plane = torch.from_numpy(x_plane(cv, wv, ntotal, ix))
        f[:, h + ix - x0, h:-h, h:-h] = plane.to(f.device)
    torch.cuda.synchronize(f.device)
