"""The log comparison of tools/regression_sweep.py (the checker of the
regression sweep and of tests/test_gpu_shim.py's hand-back test), on the logs
the reference keeps (the cases under tests/golden/regression_d3q19_short/): it must call equal
what the reference's tests/test-diff.sh + awk-fp-diff.sh call equal (same
words, printed numbers within 1e-12, volatile lines dropped) and nothing else."""

import importlib.util
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rs():
    spec = importlib.util.spec_from_file_location(
        "regression_sweep", os.path.join(HERE, "..", "tools", "regression_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _log(rs, name):
    return open(os.path.join(rs.KEPT, name + ".log")).read()


def test_every_input_has_its_log(rs):
    """The nine cases committed tests read are kept in the repository; the
    whole suite (112 of d3q19-short + 3 of d3q19-io) is staged from the
    reference by `regression_sweep.py collect` where the reference exists."""
    kept = sorted(f[:-4] for f in os.listdir(rs.KEPT) if f.endswith(".inp"))
    assert len(kept) == 9
    for n in kept:
        assert os.path.exists(os.path.join(rs.KEPT, n + ".log")), n
    if os.path.isdir(rs.STAGED):
        names = rs.names()
        assert len(names) == 115
        for n in names:
            assert os.path.exists(os.path.join(rs.DATA, n + ".log")), n
        # the eight the reference's HIP target faults on are inputs of the suite
        assert rs.REF_FAULTS <= set(names)
        # what is kept is what the reference holds
        for n in kept:
            assert _log(rs, n) == open(os.path.join(rs.DATA, n + ".log")).read()


def test_a_log_equals_itself_and_volatile_lines_do_not_count(rs):
    t = _log(rs, "serial-dist-1dp")
    assert rs.compare(t, t, 1e-12) == (0, 0.0, None)
    # other compiler, other target, the binding's own lines, other timings
    u = t.replace("[rho]", "liblbmi: a line of the binding\n[rho]", 1)
    u = re.sub(r"(?m)^(\s*Compiler:).*$", r"\1 something else", u)
    u = re.sub(r"\((\d+) calls\)", "(7 calls)", u)
    u = u + "\nHalo type: lb_halo_target (full halo)\n"
    assert rs.compare(t, u, 1e-12)[0] == 0


def test_numbers_are_compared_with_the_tolerance(rs):
    t = _log(rs, "serial-dist-1dp")
    m = re.search(r"(?m)^\[total   \]\s+(\S+)", t)
    v = float(m.group(1))
    near = t[:m.start(1)] + "%.7e" % (v + 4e-13) + t[m.end(1):]
    far = t[:m.start(1)] + "%.7e" % (v + 1e-6) + t[m.end(1):]
    assert rs.compare(t, near, 1e-12)[0] == 0 or v != 0.0   # (printed to 8 digits)
    bad, worst, first = rs.compare(t, far, 1e-12)
    assert bad == 1 and abs(worst - 1e-6) < 1e-9 and first[0].startswith("[total ]")
    assert rs.compare(t, far, 1e-5)[0] == 0


def test_words_and_structure_count(rs):
    t = _log(rs, "serial-dist-1dp")
    assert rs.compare(t, t.replace("Scalars - total mean variance min max", "Scalars - total", 1), 1e-12)[0] == 1
    lines = t.splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("[rho]"))
    gone = "\n".join(lines[:k] + lines[k + 1:])
    bad, _, first = rs.compare(t, gone, 1e-12)
    assert bad == 1 and first[0].startswith("[rho]")
    # an extra line does not shift the lines after it out of their pairs
    more = "\n".join(lines[:k] + ["[new] 1.0 2.0"] + lines[k:])
    assert rs.compare(t, more, 1e-12)[0] == 1


def test_two_different_runs_differ(rs):
    bad, worst, _ = rs.compare(_log(rs, "serial-dist-1dp"), _log(rs, "serial-wall-st1"), 1e-12)
    assert bad > 3 and worst > 1e-6
