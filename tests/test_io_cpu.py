"""Row f3, host side: the distribution files of the reference's MPI-IO mode
(lb_io_write, model.c:1568-1614) -- metadata JSON, file names and the record
stream -- against files the compiled reference wrote (tests/golden/io_*.npz).
No GPU: the metadata writer and the file-name helper are host-only entry
points of the C-ABI; the record stream is checked on the oracle."""

import os
import subprocess

import numpy as np
import pytest

from oracle import lb_oracle as lbo
from tests.common import interior, load_io_golden

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref")
CASES = [("io_q19", 19), ("io_q27", 27), ("io_q19_2dist", 19)]


def _nlocal(g):
    return tuple(n - 2 for n in g["f0"].shape[1:])


def _ndist(g, nvel):
    return g["f0"].shape[0] // nvel


@pytest.mark.parametrize("name,nvel", CASES)
def test_metadata_file_identical_to_reference(name, nvel, tmp_path):
    import ludwig_amd
    g = load_io_golden(name)
    ludwig_amd.io_metadata_write(tmp_path, "dist", nvel * _ndist(g, nvel), _nlocal(g))
    text = open(tmp_path / "dist-metadata.001-001").read()
    assert text == g["metadata"]                  # byte for byte


@pytest.mark.parametrize("name,nvel", CASES)
def test_data_file_name(name, nvel, tmp_path):
    import ludwig_amd
    g = load_io_golden(name)
    fn = ludwig_amd.io_filename(tmp_path, "dist", g["timestep"])
    assert fn == str(tmp_path / g["datafile"])


@pytest.mark.parametrize("name,nvel", CASES)
def test_oracle_record_stream_is_the_reference_file(name, nvel):
    g = load_io_golden(name)
    p = lbo.make_param(nvel, _nlocal(g), 1)
    nd = _ndist(g, nvel)
    rec = lbo.records_pack(p, np.ascontiguousarray(g["f0"]), nd)
    assert rec.tobytes() == g["data"]
    f = np.zeros_like(g["f0"])
    lbo.records_unpack(p, f, np.frombuffer(g["data"], dtype="<f8").copy(), nd)
    assert np.array_equal(interior(f, 1), interior(g["f0"], 1))


@pytest.mark.parametrize("name,nvel", CASES)
def test_reference_reads_our_files(name, nvel, tmp_path):
    """lb_io_read of the compiled reference on a file pair produced by our
    host side (metadata by the C-ABI helper, records by the oracle)."""
    exe = os.path.join(REF, "ref_driver_d3q%d" % nvel)
    if not os.path.exists(exe):
        pytest.skip("compiled reference (oracle/_ref) not present")
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    p = lbo.make_param(nvel, n, 1)
    rng = np.random.default_rng(3)
    nd = _ndist(g, nvel)
    f = np.zeros_like(g["f0"])
    interior(f, 1)[...] = rng.random((nd * nvel,) + n)
    ludwig_amd.io_metadata_write(tmp_path, "dist", nd * nvel, n)
    with open(ludwig_amd.io_filename(tmp_path, "dist", 42), "wb") as fp:
        fp.write(lbo.records_pack(p, f, nd).tobytes())
    subprocess.run([exe, "ioread", str(tmp_path), *map(str, n), "42"]
                   + ([str(nd)] if nd != 1 else []), check=True,
                   stdout=subprocess.DEVNULL)
    back = np.fromfile(tmp_path / "readback.f.f64", dtype="<f8").reshape(f.shape)
    assert np.array_equal(interior(back, 1), interior(f, 1))


@pytest.mark.parametrize("name,ndist", [("io_q19_ascii", 1), ("io_q19_2dist_ascii", 2)])
def test_text_record_metadata_and_oracle_text(name, ndist, tmp_path):
    """distribution_io_format ascii: the metadata (MPI_CHAR x nvel*(ndist*23 +
    1)) byte for byte, and the text of the data file re-made here from the f
    it was written from: per site nvel lines, line p = f(n, p) for the ndist
    distributions as " %22.15e" (lb_write_buf_ascii, model.c:1438-1462)."""
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    ludwig_amd.io_metadata_write_fmt(tmp_path, "dist", 19, ndist, n, ascii=True)
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    f = interior(g["f0"], 1).reshape((ndist, 19) + n)
    lines = []
    for ic in range(n[0]):
        for jc in range(n[1]):
            for kc in range(n[2]):
                for p in range(19):
                    lines.append("".join(" %22.15e" % f[d, p, ic, jc, kc] for d in range(ndist)))
    assert ("\n".join(lines) + "\n").encode() == g["data"]


SINGLE = [("io_q19_single", 19), ("io_q27_2dist_single", 27)]


@pytest.mark.parametrize("name,nvel", SINGLE)
def test_single_mode_files_of_a_run_that_names_no_io_mode(name, nvel, tmp_path):
    """io_options_default() is IO_MODE_SINGLE: lb_io_write goes old-style
    (model.c:1583-1587 -> io_write_data_s, io_harness.c). Three files: the
    JSON metadata (it says "single", version 1), the text file
    dist.001-001.meta, and the data under dist-%8.8d.001-001 -- the same
    record stream as the MPI-IO mode writes. The first two and the name are
    host-only entry points of the C-ABI; byte for byte against what the
    compiled reference wrote."""
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    nd = _ndist(g, nvel)
    ludwig_amd.io_metadata_write_fmt(tmp_path, "dist", nvel, nd, n, single=True)
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]
    ludwig_amd.io_single_metadata_write(tmp_path, "dist", nvel, nd, n)
    assert open(tmp_path / "dist.001-001.meta").read() == g["meta_text"]
    assert ludwig_amd.io_filename(tmp_path, "dist", g["timestep"], single=True) \
        == str(tmp_path / g["datafile"])
    p = lbo.make_param(nvel, n, 1)
    assert lbo.records_pack(p, np.ascontiguousarray(g["f0"]), nd).tobytes() == g["data"]


def test_single_mode_metadata_lists_the_slabs_in_rank_order(tmp_path):
    """More than one rank: one line per rank (rank, Cartesian coordinates,
    nlocal, offset) as io_write_metadata_file prints them (io_harness.c:410-415).
    No serial run of the reference can write this: the lines are checked
    against that format, not against a file of the reference (unpinned)."""
    import ludwig_amd
    ludwig_amd.io_single_metadata_write(tmp_path, "dist", 19, 1, (10, 4, 6), cartdim=0,
                                        nslab=[3, 4, 3])
    lines = open(tmp_path / "dist.001-001.meta").read().splitlines()
    assert lines[4] == "Number of processors:            3"
    assert lines[5] == "Cartesian communicator topology: 3 1 1"
    assert lines[12:] == ["%3d %3d %3d %3d %d %d %d %d %d %d" % (r, r, 0, 0, nx, 4, 6, off, 0, 0)
                          for r, nx, off in ((0, 3, 0), (1, 4, 3), (2, 3, 7))]
    with pytest.raises(Exception):
        ludwig_amd.io_single_metadata_write(tmp_path, "dist", 19, 1, (10, 4, 6), nslab=[3, 3, 3])


@pytest.mark.parametrize("name,nvel", SINGLE)
def test_reference_reads_our_single_mode_file(name, nvel, tmp_path):
    """lb_io_read of the compiled reference in its default mode
    (io_read_data, single_file_read: a seek per row to its global position)
    on a record stream made here."""
    exe = os.path.join(REF, "ref_driver_d3q%d" % nvel)
    if not os.path.exists(exe):
        pytest.skip("compiled reference (oracle/_ref) not present")
    import ludwig_amd
    g = load_io_golden(name)
    n = _nlocal(g)
    nd = _ndist(g, nvel)
    p = lbo.make_param(nvel, n, 1)
    rng = np.random.default_rng(5)
    f = np.zeros_like(g["f0"])
    interior(f, 1)[...] = rng.random((nd * nvel,) + n)
    with open(ludwig_amd.io_filename(tmp_path, "dist", 9, single=True), "wb") as fp:
        fp.write(lbo.records_pack(p, f, nd).tobytes())
    subprocess.run([exe, "ioread", str(tmp_path), *map(str, n), "9", str(nd), "single"],
                   check=True, stdout=subprocess.DEVNULL)
    back = np.fromfile(tmp_path / "readback.f.f64", dtype="<f8").reshape(f.shape)
    assert np.array_equal(interior(back, 1), interior(f, 1))


def test_metadata_of_a_lattice_that_is_not_periodic(tmp_path):
    """cs_to_json prints the periodicity: a run with walls in z writes
    [1, 1, 0] (fixture from the compiled reference)."""
    import ludwig_amd
    g = load_io_golden("io_q19_wallz")
    ludwig_amd.io_metadata_write_file(tmp_path, "dist", 19, 1, _nlocal(g), periodic=(1, 1, 0))
    assert open(tmp_path / "dist-metadata.001-001").read() == g["metadata"]


def test_metadata_of_one_file_of_several(tmp_path):
    """distribution_io_grid 3_1_1 over slabs: file 2 of 3 holding planes
    4..7 of 10. The keys io_subfile_to_json / io_options_to_json print
    (io_subfile.c:102-130) with the values io_subfile_create computes
    (:49-91); everything else as in the one-file metadata, which is pinned.
    No serial run of the reference writes several files: unpinned."""
    import json
    import ludwig_amd
    n = (10, 4, 6)
    ludwig_amd.io_metadata_write_file(tmp_path, "dist", 19, 1, n)
    one = json.load(open(tmp_path / "dist-metadata.001-001"))
    ludwig_amd.io_metadata_write_file(tmp_path, "dist", 19, 1, n, nfile=3, index=1,
                                      file_nx=4, file_x0=3)
    text = open(tmp_path / "dist-metadata.002-003").read()
    two = json.loads(text)
    assert two["io_subfile"] == {"Number of files": 3, "File index": 1, "Topology": [3, 1, 1],
                                 "Coordinate": [1, 0, 0], "Data ndims": 3,
                                 "File size (sites)": [4, 4, 6], "File offset (sites)": [3, 0, 0]}
    assert two["io_options"]["I/O grid"] == [3, 1, 1]
    two["io_subfile"] = one["io_subfile"]
    two["io_options"]["I/O grid"] = [1, 1, 1]
    assert two == one
    assert list(two) == list(one) and text.startswith('{\n\t"coords":\t{\n')
    with pytest.raises(Exception):       # planes outside the lattice
        ludwig_amd.io_metadata_write_file(tmp_path, "dist", 19, 1, n, nfile=3, index=2,
                                          file_nx=4, file_x0=8)
    with pytest.raises(Exception):       # several old-style files are another mode
        ludwig_amd.io_metadata_write_file(tmp_path, "dist", 19, 1, n, nfile=2, index=0,
                                          file_nx=5, file_x0=0, single=True)
