"""Row f3 on the device: lbmi_lb_io_write / lbmi_lb_io_read against files the
compiled reference wrote (tests/golden/io_*.npz), and -- where oracle/_ref
travelled to this box -- the reference reading files written here."""

import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.common import interior, load_io_golden          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref")
CASES = [("io_q19", 19), ("io_q27", 27)]


def _nlocal(g):
    return tuple(n - 2 for n in g["f0"].shape[1:])


@pytest.mark.parametrize("mode", [0, 3], ids=["eager", "fused_halo"])
def test_two_distributions_files(mode, tmp_path):
    """ndist = 2: records of 2*nvel doubles, [n][p] per site. Write = the
    reference's files byte for byte; read back; record stream round trip; and
    the compiled reference reads what was written here."""
    import ludwig_amd
    g = load_io_golden("io_q19_2dist")
    n = _nlocal(g)
    lb = ludwig_amd.LB(19, n, 1, ndist=2, mode=mode)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_io_write(tmp_path, g["timestep"])
    lb.synchronize()
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    rec = lb.lb_io_aggr_pack()
    assert rec.shape == n + (38,) and rec.tobytes() == g["data"]
    lb.lb_memcpy_h2d(np.zeros_like(g["f0"]))
    lb.lb_io_read(tmp_path, g["timestep"])
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1), interior(g["f0"], 1))
    lb.lb_memcpy_h2d(np.zeros_like(g["f0"]))
    lb.lb_io_aggr_unpack(rec)
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1), interior(g["f0"], 1))
    lb.free()
    exe = os.path.join(REF, "ref_driver_d3q19")
    if os.path.exists(exe):
        subprocess.run([exe, "ioread", str(tmp_path), *map(str, n),
                        str(g["timestep"]), "2"], check=True, stdout=subprocess.DEVNULL)
        back = np.fromfile(tmp_path / "readback.f.f64", dtype="<f8").reshape(g["f0"].shape)
        assert np.array_equal(interior(back, 1), interior(g["f0"], 1))


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["eager", "fused", "inplace"])
@pytest.mark.parametrize("name,nvel", CASES)
def test_write_identical_to_reference_files(name, nvel, mode, tmp_path):
    import ludwig_amd
    g = load_io_golden(name)
    lb = ludwig_amd.LB(nvel, _nlocal(g), 1, mode=mode)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_io_write(tmp_path, g["timestep"])
    lb.synchronize()
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    # the state is untouched by writing
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1), interior(g["f0"], 1))
    lb.free()


@pytest.mark.parametrize("name,nvel", CASES)
def test_read_reference_file(name, nvel, tmp_path):
    import ludwig_amd
    g = load_io_golden(name)
    with open(tmp_path / g["datafile"], "wb") as fp:
        fp.write(g["data"])
    lb = ludwig_amd.LB(nvel, _nlocal(g), 1, mode=ludwig_amd.FUSED)
    hy = ludwig_amd.Hydro(lb.nall, lb.device)
    lb.relaxation_set("bgk", 0.1, 0.1)
    lb.step(hy)                                # leave something pending
    lb.lb_io_read(tmp_path, g["timestep"])
    assert lb.state() == (0, 0, 0)             # reading replaces the state
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1), interior(g["f0"], 1))
    with pytest.raises(ludwig_amd.LbmiError):
        lb.lb_io_read(tmp_path, g["timestep"] + 1)       # no such file
    lb.free()


def test_restart_continues_bitwise(tmp_path):
    """Write in the middle of a FUSED run (deferred, blocked state), read
    into a fresh EAGER handle, continue both: same distributions."""
    import ludwig_amd
    from oracle import lb_oracle as lbo
    nlocal = (20, 12, 16)
    p = lbo.make_param(19, nlocal, 1, "m10", 0.1, 0.3)
    f0 = lbo.init_synthetic(p)
    a = ludwig_amd.LB(19, nlocal, 1, mode=ludwig_amd.FUSED)
    a.relaxation_set("m10", 0.1, 0.3)
    ha = ludwig_amd.Hydro(a.nall, a.device)
    a.lb_memcpy_h2d(f0)
    for _ in range(3):
        a.step(ha)
    assert a.state()[2] == 1
    a.lb_io_write(tmp_path, 3)
    b = ludwig_amd.LB(19, nlocal, 1, mode=ludwig_amd.EAGER)
    b.relaxation_set("m10", 0.1, 0.3)
    hb = ludwig_amd.Hydro(b.nall, b.device)
    b.lb_io_read(tmp_path, 3)
    for _ in range(3):
        a.step(ha)
        b.step(hb)
    assert np.array_equal(interior(a.lb_memcpy_d2h(), 1), interior(b.lb_memcpy_d2h(), 1))
    a.free()
    b.free()


def test_slabs_write_one_file(tmp_path):
    """Two X slabs write their byte ranges of ONE file; a third handle reads
    the whole, and each slab reads its part back."""
    import ludwig_amd
    g = load_io_golden("io_q19")
    n = _nlocal(g)
    cuts = [(0, 2), (2, 6)]
    for x0, x1 in reversed(cuts):              # order must not matter
        lb = ludwig_amd.LB(19, (x1 - x0, n[1], n[2]), 1)
        f = np.zeros((19, x1 - x0 + 2, n[1] + 2, n[2] + 2))
        interior(f, 1)[...] = interior(g["f0"], 1)[:, x0:x1]
        lb.lb_memcpy_h2d(f)
        lb.lb_io_write(tmp_path, 7, ntotal_x=n[0], offset_x=x0)
        lb.free()
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    x0, x1 = cuts[1]
    lb = ludwig_amd.LB(19, (x1 - x0, n[1], n[2]), 1)
    lb.lb_io_read(tmp_path, 7, ntotal_x=n[0], offset_x=x0)
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1),
                          interior(g["f0"], 1)[:, x0:x1])
    with pytest.raises(ludwig_amd.LbmiError):
        lb.lb_io_write(tmp_path, 7, ntotal_x=n[0], offset_x=x0 + 1)   # outside
    lb.free()


@pytest.mark.parametrize("name,nvel", CASES)
def test_reference_reads_files_written_here(name, nvel, tmp_path):
    exe = os.path.join(REF, "ref_driver_d3q%d" % nvel)
    if not os.path.exists(exe):
        pytest.fail("compiled reference (oracle/_ref) is missing: `make -C oracle ref` "
                    "in the development container")
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    rng = np.random.default_rng(11)
    f = np.zeros_like(g["f0"])
    interior(f, 1)[...] = rng.random((nvel,) + n)
    lb = ludwig_amd.LB(nvel, n, 1)
    lb.lb_memcpy_h2d(f)
    lb.lb_io_write(tmp_path, 42)
    lb.free()
    subprocess.run([exe, "ioread", str(tmp_path), *map(str, n), "42"], check=True,
                   stdout=subprocess.DEVNULL)
    back = np.fromfile(tmp_path / "readback.f.f64", dtype="<f8").reshape(f.shape)
    assert np.array_equal(interior(back, 1), interior(f, 1))


def test_large_file_chunks(tmp_path):
    """More than one staging chunk (32 MiB): 96^3 D3Q19 = 134 MB."""
    import ludwig_amd
    import torch
    n = (96, 96, 96)
    lb = ludwig_amd.LB(19, n, 1)
    g = torch.Generator(device=lb.device)
    g.manual_seed(5)
    lb.f[:, 1:-1, 1:-1, 1:-1] = torch.rand((19,) + n, dtype=torch.float64,
                                           device=lb.device, generator=g)
    ref = lb.f[:, 1:-1, 1:-1, 1:-1].clone()
    lb.lb_io_write(tmp_path, 1)
    assert os.path.getsize(tmp_path / "dist-000000001.001-001") == 19 * 8 * 96 ** 3
    rec = np.fromfile(tmp_path / "dist-000000001.001-001", dtype="<f8")
    assert np.array_equal(rec.reshape(n + (19,)),
                          ref.permute(1, 2, 3, 0).cpu().numpy())
    lb.f.zero_()
    lb.lb_io_read(tmp_path, 1)
    lb.synchronize()
    assert torch.equal(lb.f[:, 1:-1, 1:-1, 1:-1], ref)
    lb.free()


@pytest.mark.parametrize("name,ndist", [("io_q19_ascii", 1), ("io_q19_2dist_ascii", 2)])
def test_text_records_identical_to_reference_files(name, ndist, tmp_path):
    """distribution_io_format ascii (lb_write_buf_ascii / lb_read_buf_ascii,
    model.c:1438-1490): per site nvel lines of ndist values " %22.15e". The
    records are packed on the device, the text is made on the host; the data
    file and the metadata (MPI_CHAR x nvel*(ndist*23 + 1)) are the compiled
    reference's byte for byte, and reading them back gives the state to the
    sixteen digits the text holds. Slabs write their own character ranges of
    the one file."""
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    lb = ludwig_amd.LB(19, n, 1, ndist=ndist, mode=ludwig_amd.FUSED)
    lb.io_format_set(True)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_io_write(tmp_path, g["timestep"])
    lb.synchronize()
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    lb.free()
    # the reference's file read back
    rd = tmp_path / "rd"
    rd.mkdir()
    with open(rd / g["datafile"], "wb") as fp:
        fp.write(g["data"])
    lb = ludwig_amd.LB(19, n, 1, ndist=ndist)
    lb.io_format_set(True)
    lb.lb_io_read(rd, g["timestep"])
    back = interior(lb.lb_memcpy_d2h(), 1)
    ref = interior(g["f0"], 1)
    assert np.max(np.abs(back - ref)) <= 1e-15 * np.max(np.abs(ref))
    lb.free()
    # two slabs, each its own range of the one file
    if ndist == 1 and n[0] % 2 == 0:
        sl = tmp_path / "slabs"
        sl.mkdir()
        half = n[0] // 2
        for r in (1, 0):
            lb = ludwig_amd.LB(19, (half, n[1], n[2]), 1)
            lb.io_format_set(True)
            lb.lb_memcpy_h2d(np.ascontiguousarray(g["f0"][:, r * half:r * half + half + 2]))
            lb.lb_io_write(sl, g["timestep"], ntotal_x=n[0], offset_x=r * half)
            lb.synchronize()
            lb.free()
        assert open(sl / g["datafile"], "rb").read() == g["data"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,nvel", [("io_q19_single", 19), ("io_q27_2dist_single", 27)])
def test_single_mode_files(name, nvel, tmp_path):
    """The old-style i/o of a run that names no i/o mode (io_options_default():
    single; io_harness.c): dist-%8.8d.001-001, dist.001-001.meta and the JSON
    metadata that says "single" -- all three byte for byte what the compiled
    reference wrote; read back exactly; two slabs write their ranges of the
    one file; text records are refused in this mode (the reference gives them
    no size)."""
    import ludwig_amd
    g = load_io_golden(name)
    n = tuple(m - 2 for m in g["f0"].shape[1:])
    ndist = g["f0"].shape[0] // nvel
    lb = ludwig_amd.LB(nvel, n, 1, ndist=ndist, mode=ludwig_amd.FUSED)
    lb.io_format_set(single=True)
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_io_write(tmp_path, g["timestep"])
    lb.synchronize()
    assert sorted(os.listdir(tmp_path)) == sorted(["dist-metadata.001-001", "dist.001-001.meta",
                                                   g["datafile"]])
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    assert open(tmp_path / "dist.001-001.meta").read() == g["meta_text"]
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    with pytest.raises(Exception):
        lb.io_format_set(ascii=True, single=True)
    lb.free()
    lb = ludwig_amd.LB(nvel, n, 1, ndist=ndist)
    lb.io_format_set(single=True)
    lb.lb_io_read(tmp_path, g["timestep"])
    assert np.array_equal(interior(lb.lb_memcpy_d2h(), 1), interior(g["f0"], 1))
    lb.free()
    if n[0] % 2 == 0:
        sl = tmp_path / "slabs"
        sl.mkdir()
        half = n[0] // 2
        for r in (1, 0):
            lb = ludwig_amd.LB(nvel, (half, n[1], n[2]), 1, ndist=ndist, cartsz=2, cartrank=r)
            lb.io_format_set(single=True)
            lb.lb_memcpy_h2d(np.ascontiguousarray(g["f0"][:, r * half:r * half + half + 2]))
            lb.lb_io_write(sl, g["timestep"], ntotal_x=n[0], offset_x=r * half)
            lb.synchronize()
            lb.free()
        assert open(sl / g["datafile"], "rb").read() == g["data"]
        meta = open(sl / "dist.001-001.meta").read().splitlines()
        assert meta[4] == "Number of processors:            2"
        assert meta[12:] == ["%3d %3d %3d %3d %d %d %d %d %d %d" % (r, r, 0, 0, half, n[1], n[2],
                                                                     r * half, 0, 0) for r in (0, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("ascii_", [False, True], ids=["binary", "ascii"])
def test_several_files_along_the_slabs(ascii_, tmp_path):
    """distribution_io_grid 2_1_1 over four X slabs (io_subfile_create,
    io_subfile.c:49-91): slabs 0, 1 write file 1 of 2, slabs 2, 3 file 2 of 2,
    each its byte range at its planes' position in the FILE
    (io_impl_mpio.c:179-272). The two files one after the other are the
    one-file record stream of the whole lattice (which is pinned by the
    reference's files); each has its metadata; reading them back into the
    slabs restores the state; planes outside the file named are refused."""
    import json
    import ludwig_amd
    rng = np.random.default_rng(11)
    n = (8, 5, 4)
    nslab = [2, 2, 3, 1]                # (cs_init allows unequal parts)
    f = np.zeros((19, n[0] + 2, n[1] + 2, n[2] + 2))
    interior(f, 1)[...] = rng.random((19,) + n)
    whole = tmp_path / "whole"
    whole.mkdir()
    lb = ludwig_amd.LB(19, n, 1)
    lb.io_format_set(ascii_)
    lb.lb_memcpy_h2d(f)
    lb.lb_io_write(whole, 5)
    lb.synchronize()
    lb.free()
    stream = open(whole / "dist-000000005.001-001", "rb").read()

    x0 = [0, 2, 4, 7]
    files = [(0, 4, 0), (0, 4, 0), (1, 4, 4), (1, 4, 4)]        # index, planes, first plane
    for r in (3, 1, 0, 2):
        lb = ludwig_amd.LB(19, (nslab[r], n[1], n[2]), 1, cartsz=4, cartrank=r)
        lb.io_format_set(ascii_)
        lb.io_file_set(2, files[r][0], files[r][1], files[r][2])
        lb.lb_memcpy_h2d(np.ascontiguousarray(f[:, x0[r]:x0[r] + nslab[r] + 2]))
        lb.lb_io_write(tmp_path, 5, ntotal_x=n[0], offset_x=x0[r])
        lb.synchronize()
        if r == 3:
            with pytest.raises(Exception):
                lb.io_file_set(2, 0, 4, 0)             # plane 7 is not in file 1
                lb.lb_io_write(tmp_path, 5, ntotal_x=n[0], offset_x=x0[r])
        lb.free()
    one = open(tmp_path / "dist-000000005.001-002", "rb").read()
    two = open(tmp_path / "dist-000000005.002-002", "rb").read()
    assert one + two == stream and len(one) == len(two)
    for i in (0, 1):
        m = json.load(open(tmp_path / ("dist-metadata.%3.3d-002" % (i + 1))))
        assert m["io_subfile"]["File index"] == i and m["io_subfile"]["Number of files"] == 2
        assert m["io_subfile"]["File size (sites)"] == [4, 5, 4]
        assert m["io_subfile"]["File offset (sites)"] == [4 * i, 0, 0]
    for r in range(4):
        lb = ludwig_amd.LB(19, (nslab[r], n[1], n[2]), 1, cartsz=4, cartrank=r)
        lb.io_format_set(ascii_)
        lb.io_file_set(2, files[r][0], files[r][1], files[r][2])
        lb.lb_io_read(tmp_path, 5, ntotal_x=n[0], offset_x=x0[r])
        back = interior(lb.lb_memcpy_d2h(), 1)
        ref = f[:, x0[r] + 1:x0[r] + 1 + nslab[r], 1:-1, 1:-1]
        if ascii_:
            assert np.max(np.abs(back - ref)) <= 1e-15
        else:
            assert np.array_equal(back, ref)
        lb.free()


@pytest.mark.gpu
def test_metadata_says_what_is_periodic(tmp_path):
    """A run with walls in z: lb_io_write prints [1, 1, 0] (fixture from the
    compiled reference with cs_periodicity_set)."""
    import ludwig_amd
    g = load_io_golden("io_q19_wallz")
    n = _nlocal(g)
    lb = ludwig_amd.LB(19, n, 1)
    lb.io_file_set(1, 0, n[0], 0, periodic=(1, 1, 0))
    lb.lb_memcpy_h2d(g["f0"])
    lb.lb_io_write(tmp_path, g["timestep"])
    lb.synchronize()
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    assert open(tmp_path / g["datafile"], "rb").read() == g["data"]
    lb.io_file_set(None)
    lb.lb_io_write(tmp_path, g["timestep"])
    assert "[1, 1, 1]" in open(tmp_path / "dist-metadata.001-001").read()
    lb.free()
