"""The C-ABI driven from plain C99 (gcc), with no Python/torch in the
process: tests/c/test_abi.c reproduces the conserved quantities of the
reference's regression case serial-dist-3du in all three execution modes."""

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_c_test(tmpdir):
    exe = os.path.join(tmpdir, "test_abi")
    libdir = os.path.join(ROOT, "ludwig_amd")
    subprocess.run(
        ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I/opt/rocm/include",
         "-I" + os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tests", "c", "test_abi.c"), "-o", exe,
         "-L" + libdir, "-llbmi", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
         "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_c_program_compiles_and_links(tmp_path):
    """CPU: ANSI C code compiles against include/lbmi.h and links with
    liblbmi.so (no GPU needed for this)."""
    assert os.path.exists(build_c_test(str(tmp_path)))


@pytest.mark.gpu
def test_c_program_runs(tmp_path):
    exe = build_c_test(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C-ABI test passed" in r.stdout
