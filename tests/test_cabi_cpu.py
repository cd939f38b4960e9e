"""CPU-only checks of the C-ABI library: it loads, exports every symbol that
include/lbmi.h declares, and refuses to compute without a device."""

import ctypes
import os
import re

import numpy as np
import pytest

import ludwig_amd
from ludwig_amd import lib as L
from oracle import lb_oracle as lbo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    return L.library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbmi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "missing symbol " + name
    # and the python prototypes cover the header
    assert sorted(n for n, _, _ in L.SYMBOLS) == names


def test_options_default(lib):
    o = L.Options()
    assert lib.lbmi_options_default(ctypes.byref(o)) == 0
    assert (o.nvel, o.ndist, o.nhalo, o.cartsz) == (19, 1, 1, 1)
    assert lib.lbmi_options_default(None) < 0


@pytest.mark.parametrize("nvel", [19, 27])
def test_model_tables_match_oracle(lib, nvel):
    # reference tests/unit/test_lb_model.c:103-375, via the oracle
    m = ludwig_amd.lb.model(nvel)
    o = lbo.model(nvel)
    assert np.array_equal(m["cv"], o["cv"])
    assert np.array_equal(m["wv"], o["wv"])
    assert np.max(np.abs(m["ma"] - o["ma"])) == 0.0
    assert np.max(np.abs(m["na"] - o["na"]) / o["na"]) < 1e-15
    # cv[p] = -cv[nvel - p]
    for p in range(1, nvel):
        assert np.array_equal(m["cv"][p], -m["cv"][nvel - p])


def test_model_rejects_other_sets(lib):
    with pytest.raises(L.LbmiError):
        ludwig_amd.lb.model(15)


def test_create_argument_errors(lib):
    o = L.Options()
    h = ctypes.c_void_p()
    lib.lbmi_options_default(ctypes.byref(o))
    o.nvel = 15
    assert lib.lbmi_create(ctypes.byref(o), ctypes.byref(h)) == -2
    assert b"d3q19" in lib.lbmi_last_error()
    lib.lbmi_options_default(ctypes.byref(o))
    o.ndist = 3
    assert lib.lbmi_create(ctypes.byref(o), ctypes.byref(h)) == -2
    lib.lbmi_options_default(ctypes.byref(o))
    o.ndist = 2                      # two distributions: not the in-place mode
    o.mode = 2
    assert lib.lbmi_create(ctypes.byref(o), ctypes.byref(h)) == -2
    assert b"EAGER" in lib.lbmi_last_error()
    lib.lbmi_options_default(ctypes.byref(o))
    o.cartrank = 3
    assert lib.lbmi_create(ctypes.byref(o), ctypes.byref(h)) == -1


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    o = L.Options()
    h = ctypes.c_void_p()
    lib.lbmi_options_default(ctypes.byref(o))
    rc = lib.lbmi_create(ctypes.byref(o), ctypes.byref(h))
    assert rc == -3
    assert b"no CPU fallback" in lib.lbmi_last_error()
    with pytest.raises(L.LbmiError):
        ludwig_amd.LB(19, (4, 4, 4))


def test_slab_decomposition():
    d = ludwig_amd.SlabDecomposition((256, 256, 256), 8, 3)
    assert d.nlocal == (32, 256, 256)
    assert d.noffset == (96, 0, 0)
    assert (d.prev, d.next) == (2, 4)
    assert d.plane_doubles(5) * 8 == 258 * 258 * 5 * 8
    m = lbo.model(19)
    lo, hi = d.reduced_populations(m["cv"])
    assert len(lo) == len(hi) == 5          # model.c:1192-1219: 5 of 19
    m = lbo.model(27)
    lo, hi = d.reduced_populations(m["cv"])
    assert len(lo) == len(hi) == 9
    with pytest.raises(ValueError):
        ludwig_amd.SlabDecomposition((100, 8, 8), 8, 0)


def test_shim_compiles_against_reference():
    """integration/ludwig_shim.c is valid C against the reference headers
    (lb_t, hydro_t, map_t, ... as they are in zazu29/ludwig v0.20.1)."""
    import __graft_entry__ as g
    if not g.check_shim():
        pytest.skip("/root/reference not present on this machine")
